#!/bin/bash
# Regenerates the rocprofv3 summaries kept under profiles/ (run on the GPU box through gpurun; outputs go to
# gpurun_out/profiles_new/ and are copied into profiles/ by hand).  Counters in their own passes, never with
# a sys/hip/hsa trace.  TAG = round prefix of the file names (default r02).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r05}
export BENCH_CONFIG3=0 BENCH_VARIANT=0          # the 200-video jobs and the opt-in pipe's timing are not part of the profiled steps
O=$R/gpurun_out/profiles_new
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# kernel statistics: the default bench (4 batches in flight) and one batch in flight
BENCH_PLAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4 -- python3 $R/bench.py --steps 60 --warmup 5 --cpu-sample 0 --repeats 1 > $O/p4.log 2>&1      # BENCH_PLAIN: only warm-up + timed steps run, all of them pipelined
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --pipeline 1 --repeats 1 > $O/p1.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes; BENCH_PLAIN=1 -> exactly warmup + steps = 1 + 5 steps run
export BENCH_PLAIN=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 5 --warmup 1 --pipeline 1 --cpu-sample 0 --repeats 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 5 --warmup 1 --pipeline 1 --cpu-sample 0 --repeats 1 > $O/write.log 2>&1
unset BENCH_PLAIN
# SQ counters over the network alone
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/time_saliency.py > $O/sq.log 2>&1
cd $R
cp $(ls $O/p4/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_pipeline4.csv
cp $(ls $O/p1/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_pipeline1.csv
grep "^{\"metric\"" $O/p1.log | tail -1 > $O/${TAG}_bench_line_pipeline1_under_rocprof.json
python tools/pmc_traffic.py $O/fetch $O/write $O/${TAG}_pmc_traffic.json --steps 6 --note "bench.py --steps 5 --warmup 1 --pipeline 1 --cpu-sample 0 --repeats 1 with BENCH_PLAIN=1 (6 steps of 32 frames)" > /dev/null
python tools/pmc_sq_table.py $O/sq $O/${TAG}_pmc_sq_network.json > $O/${TAG}_pmc_sq_network.txt
rm -rf $O/p4 $O/p1 $O/fetch $O/write $O/sq
bash tools/trace_step.sh > $O/${TAG}_timeline_one_pass.txt 2>&1
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # bench.py reads roofline.traffic from here
unset BENCH_CONFIG3 BENCH_VARIANT
python bench.py --cpu-sample 32 > $O/${TAG}_bench_line.json 2> $O/bench.err
python bench.py --pipeline 1 --cpu-sample 0 > $O/${TAG}_bench_line_pipeline1.json 2>> $O/bench.err
tail -1 $O/${TAG}_bench_line.json | cut -c1-600
ls -la $O
