#!/bin/bash
# round 6, job m: the tail at two workgroups per CU -- TB = 512, LVL_CAP 3072, TP_CAP 2048, 78 KB LDS per launch, every cluster the budget holds
mkdir -p gpurun_out
O=gpurun_out/r06_m.txt
: > $O
L=$PWD/retargetvid_amd/libsvc_hip_tb512s.so
SVC_LIB=$L timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tail or cluster or blend or argsort or points or maximum or golden or beyond" --deselect tests/test_gpu_parity.py::test_largest_geometries_up_to_65025_points_equal_the_round2_kernels 2>&1 | tail -3 >> $O
SVC_LIB=$L timeout 600 python tools/soak_tail.py 150 21 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
BENCH_ARGS="--repeats 7" BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 1200 bash tools/ab_bench_libs.sh 3 libsvc_hip.so libsvc_hip_tb512s.so 2>&1 | grep -v amdgpu.ids >> $O
for L2 in libsvc_hip.so libsvc_hip_tb512s.so; do
  SVC_LIB=$PWD/retargetvid_amd/$L2 BENCH_CONFIG3_EXTRA=0 BENCH_VARIANT=0 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | L2=$L2 python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['L2'], 'driver flags: %.1f frames/s %.4f ms; tail %s; config3 %s' % (d['value'], d['ms_per_step'], {k: v for k, v in d['roofline']['tail'].items() if k.endswith('_ms')}, d['config']['config3']['seconds_all_runs']))" >> $O
done
cat $O
