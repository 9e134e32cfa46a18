#!/bin/bash
# round 6, job g: k_dwpw<NT <= 2> with the chunk's first weight step requested in front of the depthwise phase (A/B against -DDWPW_WPRE_NT=0)
mkdir -p gpurun_out
O=gpurun_out/r06_g.txt
: > $O
for i in 1 2; do
  for L in libsvc_hip.so libsvc_hip_nowpre.so; do
    echo "== $L" >> $O
    SVC_LIB=$PWD/retargetvid_amd/$L timeout 300 python tools/time_knobs.py 4 2>&1 | grep -v amdgpu.ids >> $O
  done
done
BENCH_ARGS="--repeats 7" BENCH_CONFIG3=0 BENCH_VARIANT=0 timeout 900 bash tools/ab_bench_libs.sh 2 libsvc_hip.so libsvc_hip_nowpre.so 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "families or lane_order or dwpw or taps or golden" 2>&1 | tail -2 >> $O
cat $O
