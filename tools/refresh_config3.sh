#!/bin/bash
# config 3 record for profiles/: three runs with one process (four worker streams) and three with four processes sharing the GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${TAG:-r03}
O=gpurun_out/${TAG}_config3.json
python tools/run_config3.py --videos 40 > /dev/null 2>&1
echo '{"runs": [' > $O
for i in 1 2 3; do
  python tools/run_config3.py --workers 4 2>/dev/null | tail -1 >> $O; echo ',' >> $O
done
for i in 1 2 3; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 tools/run_config3.py --ranks-per-gpu 4 --workers 1 2>/dev/null | tail -1 >> $O
  [ $i = 3 ] || echo ',' >> $O
done
echo '], "note": "tools/run_config3.py: 200 synthetic videos with the RetargetVid frame counts, 1:3 and 3:1, one MI355X; first three: one process, four worker streams; last three: four processes sharing the GPU (--ranks-per-gpu 4 --workers 1, boxes gathered over gloo)"}' >> $O
python -c "import json; d=json.load(open('$O')); print([(r['world'], r['seconds_slowest_rank'], r['video_frames_per_s_job']) for r in d['runs']])"
