#!/bin/bash
# rocprofv3 passes of the default bench command on the GPU box (run through gpurun):
#   bash tools/profile_round.sh r01e
# kernel trace + stats first, then the PMC counters in passes of their own (never together
# with a trace domain other than --kernel-trace), everything under gpurun_out/prof_<tag>/.
set -e
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- $CMD > $O/trace.log 2>&1
echo trace done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $CMD > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $CMD > $O/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/sq --output-format csv -- $CMD > $O/sq.log 2>&1
echo sq done
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq2 --output-format csv -- $CMD > $O/sq2.log 2>&1
echo sq2 done
tail -1 $O/trace.log
