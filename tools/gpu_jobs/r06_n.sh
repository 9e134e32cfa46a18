#!/bin/bash
# round 6, job n: planner threads / look-ahead of the job with shot detection inside, now that TransNet computes the kept rows only
mkdir -p gpurun_out
timeout 1500 python tools/time_shot_job_planners.py 3 2 4 3:8 3:40 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_n.txt
cat gpurun_out/r06_n.txt
