#!/bin/bash
# Does the backward sweep's slow mode (a quarter of the processes, +10 %) come with more address-translation misses?
# Several processes of tools/bwd_alloc_ab.py under rocprofv3 with the TCP's UTCL1 counters (kernel trace only);
# tools/bwd_tlb_summary.py prints duration and counters of env_bwd per process.   gpurun -- 'bash tools/bwd_tlb_probe.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tlb
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE \
      -d $O/run$i --output-format csv -- python3 $R/tools/bwd_alloc_ab.py separate > $O/run$i.log 2>&1
  tail -1 $O/run$i.log | cut -c1-80
done
python3 $R/tools/bwd_tlb_summary.py $O
