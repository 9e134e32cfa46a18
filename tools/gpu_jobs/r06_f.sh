#!/bin/bash
# round 6, job f: the rocprofv3 summaries kept under profiles/ (r06_*), configs 4 / 5, TransNet kernel stats
mkdir -p gpurun_out
TAG=r06 timeout 1500 bash tools/refresh_profiles.sh > gpurun_out/r06_f_refresh.log 2>&1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python tools/run_config45.py --out gpurun_out/profiles_new/r06_config4_config5.json > gpurun_out/r06_f_config45.log 2>&1
# TransNet kernel statistics (kept rows only): 800 frames per call
O=$R/gpurun_out/transnet_stats; rm -rf $O; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && CPU=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/tools/time_transnet.py > $O/run.log 2>&1)
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) gpurun_out/profiles_new/r06_transnet_kernel_stats.csv
rm -rf $O/raw
tail -3 gpurun_out/r06_f_refresh.log
ls gpurun_out/profiles_new
