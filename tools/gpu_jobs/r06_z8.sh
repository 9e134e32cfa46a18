#!/bin/bash
# round 6, job z8: soak of the video path (TransNet inside the ingest) against the oracle pipeline fed the oracle's probabilities
mkdir -p gpurun_out
O=gpurun_out/r06_z8.txt
: > $O
timeout 420 python tools/soak_video_path.py 24 6 2>&1 | grep -v amdgpu.ids | tail -40 >> $O
cat $O
