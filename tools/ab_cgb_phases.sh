#!/bin/bash
# k_cgb phase timing: the kernel's average duration with a phase left out (libraries built with -DCGB_SKIP=<bits> under retargetvid_amd/ab/)
R=${GRAFT_REPO_ROOT:-/root/repo}
export SVC_CGB=1
cd /tmp && export TMPDIR=/tmp
for L in ab/skip7.so ab/skip15.so ab/skip23.so ab/skip31.so; do
  export SVC_LIB=$R/retargetvid_amd/$L
  O=$R/gpurun_out/cgb_$(basename $L .so); rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/time_saliency.py > /dev/null 2>&1
  f=$(ls $O/*/*kernel_stats.csv | head -1)
  echo "$L: $(grep k_cgb $f | awk -F, '{printf "%s avg %.1f us; ", substr($1,1,14), $4/1e3}')"
  rm -rf $O
done
