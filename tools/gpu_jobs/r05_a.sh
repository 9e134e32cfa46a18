#!/bin/bash
# round 5, job A: the whole GPU suite with the split-bf16 default, the bench at the driver's flags, knob sweep, TB=512 tail suite
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputest_mx.txt 2>&1; tail -5 gpurun_out/r05_gputest_mx.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_mx_a.json 2> gpurun_out/r05_bench_mx_a.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_mx_a.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["config3"].get("seconds_all_runs"))
PY
python tools/time_knobs.py 4 SVC_PWR_NT=3 SVC_PWR_NT=4 SVC_PWR_NT=0 SVC_DWPW_NT=3 SVC_DWPW_NT=4 SVC_FUSE_MAX=5 SVC_FUSE_MAX=3 SVC_FUSE_MAX=10 SVC_DWPW_MIN_PX=100 > gpurun_out/r05_mx_knobs3.txt 2>&1; cat gpurun_out/r05_mx_knobs3.txt
SVC_LIB=$PWD/retargetvid_amd/libsvc_hip_tb512.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or cluster or blend or argsort or points or maximum" > gpurun_out/r05_tb512_tail.txt 2>&1; tail -5 gpurun_out/r05_tb512_tail.txt
