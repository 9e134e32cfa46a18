#!/bin/bash
# how many backbone blocks run as the fused inverted-residual kernel: isolated network time and the pipelined bench
for f in 7 10 13; do
  SVC_FUSE_MAX=$f TAG=fuse$f python tools/time_saliency.py 2>&1 | tail -1
  SVC_FUSE_MAX=$f python bench.py --cpu-sample 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
