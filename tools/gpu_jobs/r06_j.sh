#!/bin/bash
# round 6, job j: k_dwpw with the chunk loop rotated (next chunk's taps in flight during the MFMA phase; depthwise weights in LDS): NT <= 3 / <= 2 / off
mkdir -p gpurun_out
O=gpurun_out/r06_j.txt
: > $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 >> $O
for i in 1 2; do
  for L in libsvc_hip.so libsvc_hip_rot2.so libsvc_hip_rot0.so; do
    echo "== $L" >> $O
    SVC_LIB=$PWD/retargetvid_amd/$L timeout 300 python tools/time_knobs.py 4 2>&1 | grep -v amdgpu.ids >> $O
  done
done
BENCH_ARGS="--repeats 7" BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 900 bash tools/ab_bench_libs.sh 2 libsvc_hip.so libsvc_hip_rot2.so libsvc_hip_rot0.so 2>&1 | grep -v amdgpu.ids >> $O
cat $O
