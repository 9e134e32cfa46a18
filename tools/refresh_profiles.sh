#!/bin/bash
# Regenerates the rocprofv3 summaries kept under profiles/ (run on the GPU box through gpurun; outputs go to
# gpurun_out/profiles_new/ and are copied into profiles/ by hand).  Counters in their own passes, never with
# a sys/hip/hsa trace.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_new
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4 -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $O/p4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 --pipeline 1 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --pipeline 1 --cpu-sample 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --pipeline 1 --cpu-sample 0 > $O/write.log 2>&1
cd $R
cp $(ls $O/p4/*/*kernel_stats.csv | head -1) $O/kernel_stats_pipeline4.csv
cp $(ls $O/p1/*/*kernel_stats.csv | head -1) $O/kernel_stats_pipeline1.csv
python tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json > /dev/null
rm -rf $O/p4 $O/p1 $O/fetch $O/write
cp $O/pmc_traffic.json profiles/r01_pmc_traffic.json      # bench.py reads roofline.traffic from here
python bench.py > $O/bench_line.json 2> $O/bench.err
tail -1 $O/bench_line.json | cut -c1-400
ls -la $O
