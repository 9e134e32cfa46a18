#!/bin/bash
# k_cgb phase timing: the kernel's average duration with a phase left out.  Build the variants first (build container):
#   cd retargetvid_amd/csrc && mkdir -p ../ab && for v in 7 15 23 31; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
#     -ffp-contract=off -Wno-unused-function -Wno-pass-failed -DCGB_SKIP=$v -shared -o ../ab/skip$v.so svc_net.hip svc_tail.hip svc_shot.hip; done
# CGB_SKIP bits: 1 expand loop, 2 depthwise, 4 project loop, 8 partial-sum stores, 16 input rows (results are wrong, the time is the point)
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
