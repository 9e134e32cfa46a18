#!/bin/bash
# round 6, job s: rank-parallel candidate extraction in the Prim's rounds (A/B against the loop over the words): tail suite, soaks, stamps, bench
mkdir -p gpurun_out
O=gpurun_out/r06_s.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -2 >> $O
timeout 900 python tools/soak_tail.py 300 51 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
SOAK_HW=140x250 timeout 900 python tools/soak_tail.py 40 52 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
timeout 900 python tools/soak_big_maps.py 4 7 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
for L in libsvc_hip_before.so libsvc_hip.so; do
  echo "== $L tail_vs_n" >> $O
  SVC_LIB=$PWD/retargetvid_amd/$L timeout 600 python tools/tail_vs_n.py 2>&1 | grep -v amdgpu.ids | head -4 >> $O
done
SVC_PRIM_LVL=3 SIGMA=30,44 N_BLOBS=2 timeout 300 python tools/bench_map_sizes.py 2>&1 | grep "k_prim_lvl" | head -3 >> $O
BENCH_ARGS="--repeats 7" BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 1500 bash tools/ab_bench_libs.sh 3 libsvc_hip_before.so libsvc_hip.so 2>&1 | grep -v amdgpu.ids >> $O
for L in libsvc_hip_before.so libsvc_hip.so; do
  SVC_LIB=$PWD/retargetvid_amd/$L BENCH_CONFIG3_EXTRA=0 BENCH_VARIANT=0 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | L2=$L python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['L2'], 'driver flags: %.1f frames/s %.4f ms; tail %s; config3 %s' % (d['value'], d['ms_per_step'], {k: v for k, v in d['roofline']['tail'].items() if k.endswith('_ms')}, d['config']['config3']['seconds_all_runs']))" >> $O
done
cat $O
