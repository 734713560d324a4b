#!/bin/bash
# HBM traffic of hipdsp_spectrogram per window length (rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE each in a pass of
# its own, on tools/spec_sizes_bench.py -- 64 ch x 120 s x 96 kHz, hop = nfft / 2), next to the algorithmic bytes:
#   gpurun -- 'bash tools/spec_pmc.sh 256 4096 8192 65536'   ->  gpurun_out/spec_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/spec_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export WARM_CALLS=0 TIMED_CALLS=4      # the summaries count on 2 + 4 calls per variant and no other kernel
for n in "$@"; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$n --output-format csv -- python3 $R/tools/spec_sizes_bench.py $n > $O/fetch_$n.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$n --output-format csv -- python3 $R/tools/spec_sizes_bench.py $n > $O/write_$n.log 2>&1 || exit 1
  echo "nfft $n done"
done
python3 $R/tools/spec_pmc_summary.py $O "$@" | tee $O/summary.txt
