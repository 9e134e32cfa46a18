#!/bin/bash
# smoke(), the shot-net job with 1 / 2 / 3 planner threads, a final soak of the defaults
mkdir -p gpurun_out
O=gpurun_out/r05_final_checks.txt
: > $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
for p in 1 2 3; do
  echo "== planner threads $p" >> $O
  SVC_PLANNERS=$p BENCH_SHOT_ONLY=1 python - >> $O 2>&1 <<'P'
import os, sys, json, time
sys.argv = ['bench.py']
import bench, torch
from retargetvid_amd import scheduler, weights
scheduler.JobScheduler.PLANNERS = int(os.environ['SVC_PLANNERS'])
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
sd = weights.make_synthetic_state_dict(0)
r = bench.config3_job(1, 0, False, dev, sd, 12, None, mode='shot_net')
print(r['seconds'], r['seconds_all_runs'], r['windows_crc32'], r['scheduler_rank0']['feeder_seconds'])
P
done
echo "== soak: every family on the bf16 pipe, 4 streams, defaults" >> $O
timeout 900 python tools/soak_network_concurrent.py 4 1000 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
timeout 900 python tools/soak_pipeline_concurrent.py 4 300 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
cat $O
