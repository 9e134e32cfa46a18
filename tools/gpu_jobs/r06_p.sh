#!/bin/bash
# round 6, job p: the Prim rounds on 8 / 2 worker wavefronts instead of 4 (LVL_WORKERS): tail suite, tail time, pipelined bench
mkdir -p gpurun_out
O=gpurun_out/r06_p.txt
: > $O
for L in libsvc_hip_w8.so libsvc_hip_w2.so; do
  echo "== $L" >> $O
  SVC_LIB=$PWD/retargetvid_amd/$L timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tail or cluster or blend or points or maximum or golden or beyond" 2>&1 | tail -2 >> $O
  SVC_LIB=$PWD/retargetvid_amd/$L timeout 600 python tools/soak_tail.py 100 31 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
for L in libsvc_hip.so libsvc_hip_w8.so libsvc_hip_w2.so; do
  echo "== $L tail_vs_n" >> $O
  SVC_LIB=$PWD/retargetvid_amd/$L timeout 600 python tools/tail_vs_n.py 2>&1 | grep -v amdgpu.ids | head -4 >> $O
done
BENCH_ARGS="--repeats 7" BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 1200 bash tools/ab_bench_libs.sh 2 libsvc_hip.so libsvc_hip_w8.so libsvc_hip_w2.so 2>&1 | grep -v amdgpu.ids >> $O
cat $O
