#!/bin/bash
echo "== all families, smoothing kernel isolated"; LANES=12,4 SVC_MX=bf16x6 python tools/soak_job_repeat.py 80 100 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-200
echo "== k_irb only"; LANES=12,4 SVC_MX=bf16x6 SVC_MX_MASK=2 python tools/soak_job_repeat.py 80 100 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200
