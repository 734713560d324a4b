#!/bin/bash
# The regression gate's measurement: tools/entry_points_bench.py (and tools/spec_sizes_bench.py) with a BASE build of the
# library and with the tree's build in ONE box lease, base - new - base - new, then tools/entry_points_gate.py over the
# second pair (the clocks have settled by then; the first pair is kept for the spread).
#   gpurun --timeout 900 -- 'bash tools/entry_points_ab.sh tools/_ab/libr04.so r05a'
# writes gpurun_out/<tag>_entry_points_{base,new}.log, <tag>_spec_sizes_{base,new}.log and <tag>_entry_points_gate.log;
# exit code = the gate's.  SECONDS_ shortens the traces (default: the full 600 s of BASELINE configs[2]).
cd ${GRAFT_REPO_ROOT:-/root/repo}
BASE=${1:-tools/_ab/libbase.so}
TAG=${2:-ab}
SECS=${SECONDS_:-600}
OUT=gpurun_out
mkdir -p $OUT
export AUDIAN_AMD_NO_AUTOBUILD=1
for rnd in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export AUDIAN_AMD_LIB=$PWD/$BASE; else unset AUDIAN_AMD_LIB; fi
    FACADE=${FACADE:-0} timeout -k 10 300 python tools/entry_points_bench.py $SECS > $OUT/${TAG}_entry_points_${v}_$rnd.log 2>&1 || { echo "entry_points_bench ($v) failed"; tail -5 $OUT/${TAG}_entry_points_${v}_$rnd.log; exit 3; }
    echo "entry points, $v, round $rnd: done"
  done
done
# (the window-length launches take 1 - 2 ms each: three rounds, the fastest time of a line counts)
for rnd in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export AUDIAN_AMD_LIB=$PWD/$BASE; else unset AUDIAN_AMD_LIB; fi
    TIMED_CALLS=16 timeout -k 10 300 python tools/spec_sizes_bench.py > $OUT/${TAG}_spec_sizes_${v}_$rnd.log 2>&1 || { echo "spec_sizes_bench ($v) failed"; exit 3; }
    echo "window lengths, $v, round $rnd: done"
  done
done
unset AUDIAN_AMD_LIB
cp $OUT/${TAG}_entry_points_base_2.log $OUT/${TAG}_entry_points_base.log
cp $OUT/${TAG}_entry_points_new_2.log $OUT/${TAG}_entry_points_new.log
cp $OUT/${TAG}_spec_sizes_base_3.log $OUT/${TAG}_spec_sizes_base.log
cp $OUT/${TAG}_spec_sizes_new_3.log $OUT/${TAG}_spec_sizes_new.log
{
  python tools/entry_points_gate.py $OUT/${TAG}_entry_points_base.log $OUT/${TAG}_entry_points_new.log --also-base $OUT/${TAG}_entry_points_base_1.log --also-new $OUT/${TAG}_entry_points_new_1.log "${@:3}"; rc1=$?
  python tools/entry_points_gate.py $OUT/${TAG}_spec_sizes_base.log $OUT/${TAG}_spec_sizes_new.log --also-base $OUT/${TAG}_spec_sizes_base_1.log --also-base $OUT/${TAG}_spec_sizes_base_2.log --also-new $OUT/${TAG}_spec_sizes_new_1.log --also-new $OUT/${TAG}_spec_sizes_new_2.log "${@:3}"; rc2=$?
  echo "base = $BASE ($(sha1sum $BASE | cut -c1-12)), new = audian_amd/libhip_dsp.so ($(sha1sum audian_amd/libhip_dsp.so | cut -c1-12))"
  exit $((rc1 | rc2))
} 2>&1 | tee $OUT/${TAG}_entry_points_gate.log
exit ${PIPESTATUS[0]}
