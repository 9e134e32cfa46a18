#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_transnet.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -3
python tools/time_knobs.py 4 SVC_SD_EXCL=1 2>&1 | grep -v amdgpu.ids
SECONDS=0
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_flags.json 2> gpurun_out/r05_bench_driver_flags.err; echo "bench wall seconds: $SECONDS"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_driver_flags.json").read().strip().splitlines()[-1])
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["matrix_pipe"], c["points_per_map"])
for k in ("config3", "config3_host_fed", "config3_shot_net"):
    v = c.get(k) or {}
    print(k, v.get("error"), v.get("seconds"), v.get("seconds_all_runs"), v.get("saliency_frames"), v.get("per_rank_fixed_costs_s"))
print({k: v for k, v in (c.get("matrix_pipe_variant") or {}).items() if k != 'note'})
PY
echo "== TB=512, LVL_CAP 3072, TP_CAP 2432, 78 KB LDS budget"
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_tb512s.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or cluster or blend or argsort or points or maximum" 2>&1 | tail -3
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_tb512s.so BENCH_CONFIG3_EXTRA=0 BENCH_VARIANT=0 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/r05_bench_tb512s.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_tb512s.json").read().strip().splitlines()[-1])
print("TB=512 small:", d["value"], d["ms_per_step"], {k: v for k, v in d["roofline"]["tail"].items() if k != 'note'}, d["config"]["config3"]["seconds_all_runs"], d["config"]["batch_phase_ms_in_the_pipeline"])
PY
