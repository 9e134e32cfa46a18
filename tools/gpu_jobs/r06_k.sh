#!/bin/bash
# round 6, job k: end-to-end window / IoU parity of the final build against the oracle pipeline, four checkpoint families, both parameter sets
mkdir -p gpurun_out
PARITY_CHECKPOINTS=tl,tl2,ri,nc PARITY_OUT=r06_iou_parity_all.json timeout 3000 python tools/iou_parity.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_k.txt
cat gpurun_out/r06_k.txt
