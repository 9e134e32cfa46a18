#!/bin/bash
# round 6, job z9: the video path's multi-scene bookkeeping against the oracle's, a frame-difference detector standing in for TransNet on both sides
mkdir -p gpurun_out
O=gpurun_out/r06_z9.txt
: > $O
SOAK_SHOT_NET=diff timeout 300 python tools/soak_video_path.py 60 7 2>&1 | grep -v amdgpu.ids | tail -40 >> $O
cat $O
