#!/bin/bash
# L2 (TCC) / fabric counters of the envelope's backward sweep next to the float4 copy that reaches 6.2 TB/s on the
# same box: where does the memory system push back?   gpurun -- 'bash tools/tcc_probe.sh'; tools/tcc_summary.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tcc
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 $R/tools/copy_sweep.hip -o /tmp/copy_sweep 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# (at most four TCC counters per pass: more "exceeds the capabilities of the hardware" and rocprofv3 aborts)
n=1
PASSES=${PASSES:-tcc}
if [ "$PASSES" = "tcp" ]; then
  set -- "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
         "TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum"
else
  set -- "TCC_CYCLE_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum" \
         "TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
fi
for P in "$@"; do
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $P -d $O/bwd$n --output-format csv -- python3 $R/tools/bwd_alloc_ab.py separate > $O/bwd$n.log 2>&1
  echo "pass $n: backward sweep done ($?)"
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $P -d $O/copy$n --output-format csv -- /tmp/copy_sweep > $O/copy$n.log 2>&1
  echo "pass $n: copy done ($?)"
  n=$((n+1))
done
python3 $R/tools/tcc_summary.py $O
