#!/bin/bash
# The regression gate's measurement: tools/entry_points_bench.py and tools/spec_sizes_bench.py with a BASE build of the
# library and with the tree's build in ONE PROCESS on the same device buffers (LIBS=base,tree: every entry point timed for
# both in turn, the fastest of the rounds counts), then tools/entry_points_gate.py over the two logs.
#   gpurun --timeout 900 -- 'bash tools/entry_points_ab.sh tools/_ab/libr04.so r05z'
# writes gpurun_out/<tag>_entry_points_{<base>,tree}.log, <tag>_spec_sizes_{<base>,tree}.log and <tag>_entry_points_gate.log;
# exit code = the gate's.  SECONDS_ shortens the traces (default: the full 600 s of BASELINE configs[2]); further arguments
# go to the gate (--allow 'substring=reason').
# (Until round 5 the two builds ran in processes of their own: identical machine code then differed by 5-12 % on the
# envelope's backward sweep and on the PSD kernels with the dB image -- where a process's buffers land in HBM --, which
# no tolerance of 3 % survives: profiles/r05x_entry_points_gate.log, r05y_.)
cd ${GRAFT_REPO_ROOT:-/root/repo}
BASE=${1:-tools/_ab/libbase.so}
TAG=${2:-ab}
SECS=${SECONDS_:-600}
OUT=gpurun_out
mkdir -p $OUT
BN=$(basename $BASE .so)
LIBS=$BASE,tree OUT_PREFIX=$OUT/${TAG}_entry_points timeout -k 10 600 python tools/entry_points_bench.py $SECS > $OUT/${TAG}_entry_points_run.log 2>&1 || { echo "entry_points_bench failed"; tail -5 $OUT/${TAG}_entry_points_run.log; exit 3; }
echo "entry points: done"
LIBS=$BASE,tree OUT_PREFIX=$OUT/${TAG}_spec_sizes TIMED_CALLS=16 timeout -k 10 600 python tools/spec_sizes_bench.py > $OUT/${TAG}_spec_sizes_run.log 2>&1 || { echo "spec_sizes_bench failed"; tail -5 $OUT/${TAG}_spec_sizes_run.log; exit 3; }
echo "window lengths: done"
{
  python tools/entry_points_gate.py $OUT/${TAG}_entry_points_$BN.log $OUT/${TAG}_entry_points_tree.log "${@:3}"; rc1=$?
  python tools/entry_points_gate.py $OUT/${TAG}_spec_sizes_$BN.log $OUT/${TAG}_spec_sizes_tree.log "${@:3}"; rc2=$?
  echo "base = $BASE ($(sha1sum $BASE | cut -c1-12)), new = audian_amd/libhip_dsp.so ($(sha1sum audian_amd/libhip_dsp.so | cut -c1-12)); both in one process on the same buffers"
  exit $((rc1 | rc2))
} 2>&1 | tee $OUT/${TAG}_entry_points_gate.log
exit ${PIPESTATUS[0]}
