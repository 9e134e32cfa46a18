#!/bin/bash
# round 5, job H: the final default (bf16x6 + the smoothing kernel alone on its CU): suite twice, bench, profile refresh, config 4/5
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_final1.txt 2>&1; tail -3 gpurun_out/r05_gputest_final1.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_final2.txt 2>&1; tail -3 gpurun_out/r05_gputest_final2.txt
SECONDS=0
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_flags.json 2> gpurun_out/r05_bench_driver_flags.err; echo "bench wall seconds: $SECONDS"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_driver_flags.json").read().strip().splitlines()[-1])
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["matrix_pipe"], c["points_per_map"], d["roofline"].get("frac_wall"))
for k in ("config3", "config3_host_fed", "config3_shot_net"):
    v = c.get(k) or {}
    print(k, v.get("error"), v.get("seconds"), v.get("seconds_all_runs"), v.get("saliency_frames"))
print({k: v for k, v in (c.get("matrix_pipe_variant") or {}).items() if k != 'note'})
print({k: v for k, v in (c.get("one_batch_for_every_slot") or {}).items() if k != 'note'})
PY
TAG=r05 bash tools/refresh_profiles.sh > gpurun_out/r05_refresh.log 2>&1; tail -3 gpurun_out/r05_refresh.log | cut -c1-300
python tools/run_config45.py > gpurun_out/r05_config45.log 2>&1; tail -5 gpurun_out/r05_config45.log | cut -c1-300
