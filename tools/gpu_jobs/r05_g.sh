#!/bin/bash
# round 5, job G: everything again with the split-bf16 pipe as the process-wide choice (SVC_MX=bf16x6: the smoothing kernel isolated)
mkdir -p gpurun_out
export SVC_MX=bf16x6
for k in 1 2 3; do python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_mx_iso$k.txt 2>&1; tail -3 gpurun_out/r05_gputest_mx_iso$k.txt; done
echo "== job soak"; LANES=12,4,8 python tools/soak_job_repeat.py 80 100 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200
echo "== network soak"; python tools/soak_network_concurrent.py 4 1500 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200
echo "== pipeline soak"; python tools/soak_pipeline_concurrent.py 4 600 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200
echo "== flip sources"; FLIP_OUT=r05_flip_sources_gpu_bf16x6.json python tools/flip_sources.py --gpu 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
echo "== iou parity"; PARITY_OUT=r05_iou_parity_bf16x6.json PARITY_CHECKPOINTS=tl,tl2,ri python tools/iou_parity.py 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-200
