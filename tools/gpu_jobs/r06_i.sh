#!/bin/bash
# round 6, job i: soaks of the FINAL build -- tail vs oracle, scheduler vs one-video runs (incl. the per-video census), big maps, the job with
# shot detection inside run after run (bounded look-ahead, cloned planners), the whole per-batch path with four streams
mkdir -p gpurun_out
O=gpurun_out/r06_i.txt
: > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | grep -v amdgpu.ids | tail -3 >> $O; }
run timeout 1200 python tools/soak_tail.py 400 11
SOAK_HW=140x250 run timeout 1200 python tools/soak_tail.py 60 12
run timeout 1200 python tools/soak_scheduler.py 12 5
run timeout 900 python tools/soak_big_maps.py 6 3
run timeout 1200 python tools/soak_shot_job_repeat.py 60 10
run timeout 900 python tools/soak_pipeline_concurrent.py 4 600
run timeout 900 python tools/soak_job_repeat.py 100 5
cat $O
