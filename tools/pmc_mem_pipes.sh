#!/bin/bash
# Memory-pipeline counters of the network kernels (lone passes): rocprofv3 --pmc passes over tools/time_saliency.py, a block's
# counters two at a time (more per pass: "Request exceeds the capabilities of the hardware", and the tool then hangs: every pass
# runs under its own timeout)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_mem; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_BUSY_avr" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -- python3 $R/tools/time_saliency.py > $O/p$i.log 2>&1
  python3 $R/tools/pmc_generic_table.py $O/p$i > $O/t$i.txt 2>/dev/null
  rm -rf $O/p$i
done
head -12 $O/t1.txt | cut -c1-120
