#!/bin/bash
# A/B of two BUILDS of the library on one GPU box (separate processes, alternating):
#   here:  build variant A, cp audian_amd/libhip_dsp.so tools/_ab/libbase.so; build variant B, cp ... tools/_ab/libnew.so
#   gpurun -- 'bash tools/ab_two_builds.sh'          (tools/_ab/ is git-ignored but travels to the box)
# Timing differences between boxes (5-10 %) are larger than most of what is worth measuring; this keeps both
# variants on the same box.  The command timed is tools/chain_ab.py (the fused forward sweep, configs[2]).
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp audian_amd/libhip_dsp.so /tmp/libhip_dsp.keep
for rnd in 1 2 3; do
  for v in base new; do
    cp tools/_ab/lib$v.so audian_amd/libhip_dsp.so
    echo -n "$v: "; AUDIAN_AMD_NO_AUTOBUILD=1 BITS=0 timeout -k 5 100 python tools/chain_ab.py | tail -1
  done
done
cp /tmp/libhip_dsp.keep audian_amd/libhip_dsp.so
