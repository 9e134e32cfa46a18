#!/bin/bash
# config 3 on ONE GPU with R processes sharing it (each with W worker streams): does a second interpreter help the host side?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/run_config3.py --videos 40 > /dev/null 2>&1
for cfg in "1 4" "2 2" "2 3" "2 4" "3 2" "4 2" "4 1"; do
  set -- $cfg
  if [ $1 = 1 ]; then
    python tools/run_config3.py --workers $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ranks 1 workers $2:', d['seconds_slowest_rank'], 's', d['video_frames_per_s_job'], 'video frames/s', d['eval'])"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29517 tools/run_config3.py --ranks-per-gpu $1 --workers $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ranks $1 workers $2:', d['seconds_slowest_rank'], 's', d['video_frames_per_s_job'], 'video frames/s', d['eval'])"
  fi
done
