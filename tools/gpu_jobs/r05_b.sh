#!/bin/bash
# round 5, job B: the whole GPU suite twice (no -x)
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_mx2.txt 2>&1; tail -12 gpurun_out/r05_gputest_mx2.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_mx3.txt 2>&1; tail -12 gpurun_out/r05_gputest_mx3.txt
