#!/bin/bash
# round 6, job l: re-validation after the LookAhead refactor: the scheduler / planner / pipeline / distributed tests, then the shot-net job soak
mkdir -p gpurun_out
O=gpurun_out/r06_l.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_transnet.py tests/test_gpu_scheduler.py tests/test_gpu_pipeline.py tests/test_gpu_dist_rccl.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3 >> $O
timeout 900 python tools/soak_shot_job_repeat.py 60 6 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
cat $O
