#!/bin/bash
# A/B of tuning knobs on the network alone (B=32): each line of stdin / each argument is an env assignment list
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  env TAG="$v" $v python3 tools/time_saliency.py 2>/dev/null | tail -1
done
