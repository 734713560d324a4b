#!/bin/bash
# SQ counters of chain_fwd_kernel per BUILD of the library and window: where do the instructions of two builds differ?
#   gpurun -- 'bash tools/pmc_fwd_libs.sh TAG "1024:256 2048:1024" tools/_ab/libbase.so tools/_ab/libplain.so tree'
# ("tree" = audian_amd/libhip_dsp.so).  Output: gpurun_out/<TAG>_pmc_fwd_libs.txt, one line per build, window and pass.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; SHAPES=$2; shift 2
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AUDIAN_AMD_NO_AUTOBUILD=1 CALLS=4
for lib in "$@"; do
  for sh in $SHAPES; do
    export NFFT=${sh%%:*} HOP=${sh##*:}
    if [ $lib = tree ]; then unset AUDIAN_AMD_LIB; else export AUDIAN_AMD_LIB=$R/$lib; fi
    name=$(basename $lib .so)_${NFFT}_${HOP}
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $O/$name.a --output-format csv -- python3 $R/tools/chain_fwd_only.py > $O/$name.a.log 2>&1 || { echo "pass a failed for $name"; tail -3 $O/$name.a.log; exit 3; }
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d $O/$name.b --output-format csv -- python3 $R/tools/chain_fwd_only.py > $O/$name.b.log 2>&1 || { echo "pass b failed for $name"; tail -3 $O/$name.b.log; exit 3; }
    echo "$name done"
  done
done
python3 $R/tools/pmc_fwd_summary.py $O > $R/gpurun_out/${TAG}_pmc_fwd_libs.txt
cat $R/gpurun_out/${TAG}_pmc_fwd_libs.txt
