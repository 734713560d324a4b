set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_chain; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SECONDS_=600
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/sq --output-format csv -- python3 $R/tools/chain_bench.py > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq2 --output-format csv -- python3 $R/tools/chain_bench.py > $O/sq2.log 2>&1
echo done
