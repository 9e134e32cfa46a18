#!/bin/bash
for rep in 1 2 3; do for p in 2 3 4; do
  SVC_PLANNERS=$p python - 2>/dev/null <<'P'
import os, sys
sys.argv = ['bench.py']
import bench, torch
from retargetvid_amd import scheduler, weights
scheduler.JobScheduler.PLANNERS = int(os.environ['SVC_PLANNERS'])
torch.cuda.set_device(0)
r = bench.config3_job(1, 0, False, torch.device('cuda', 0), weights.make_synthetic_state_dict(0), 12, None, mode='shot_net')
print('planners', os.environ['SVC_PLANNERS'], r['seconds'], r['seconds_all_runs'], r['windows_crc32'])
P
done; done
