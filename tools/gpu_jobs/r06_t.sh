#!/bin/bash
# round 6, job t: final state -- regenerate the r06 profile summaries, then the full validation (GPU suite, smoke, the driver's bench command)
bash tools/gpu_jobs/r06_f.sh > /dev/null 2>&1
bash tools/gpu_jobs/r06_h.sh
