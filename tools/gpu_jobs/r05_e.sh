#!/bin/bash
# round 5, job E: the bench at the driver's flags (wall time), then the profile refresh (TAG=r05)
mkdir -p gpurun_out
SECONDS=0
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_flags.json 2> gpurun_out/r05_bench_driver_flags.err; echo "bench wall seconds: $SECONDS"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_driver_flags.json").read().strip().splitlines()[-1])
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["matrix_pipe"])
for k in ("config3", "config3_host_fed", "config3_shot_net"):
    v = c.get(k) or {}
    print(k, v.get("error"), v.get("seconds"), v.get("seconds_all_runs"), v.get("saliency_frames"), v.get("per_rank_fixed_costs_s"))
print(c.get("matrix_pipe_variant"))
print(d["cpu_baseline"])
PY
TAG=r05 bash tools/refresh_profiles.sh > gpurun_out/r05_refresh.log 2>&1; tail -5 gpurun_out/r05_refresh.log
