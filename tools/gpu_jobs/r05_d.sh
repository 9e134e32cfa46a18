#!/bin/bash
# round 5, job D: the whole GPU suite twice with the final defaults, the bench at the driver's flags (wall time), the TB=512 tail build
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_final1.txt 2>&1; tail -4 gpurun_out/r05_gputest_final1.txt
python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_final2.txt 2>&1; tail -4 gpurun_out/r05_gputest_final2.txt
/usr/bin/time -v python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_flags.json 2> gpurun_out/r05_bench_driver_flags.err; grep "Elapsed" gpurun_out/r05_bench_driver_flags.err
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
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_tb512.so BENCH_CONFIG3_EXTRA=0 BENCH_VARIANT=0 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/r05_bench_tb512.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_tb512.json").read().strip().splitlines()[-1])
print("TB=512:", d["value"], d["ms_per_step"], d["roofline"]["tail"], d["config"]["config3"]["seconds_all_runs"], d["config"]["batch_phase_ms_in_the_pipeline"])
PY
