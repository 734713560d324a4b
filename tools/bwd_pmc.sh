# SQ counters of the backward envelope sweep for two builds of the library (A = the tree's, B = $1), same driver.
#   gpurun -- 'bash tools/bwd_pmc.sh tools/_ab/libregw.so'
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bwd_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SECONDS_=600 WAVES=8 PREFETCH=1
for v in A B; do
  if [ $v = B ]; then export AUDIAN_AMD_LIB=$R/$1; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/$v/sq --output-format csv -- python3 $R/tools/env_bench.py > $O/$v.sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/$v/sq2 --output-format csv -- python3 $R/tools/env_bench.py > $O/$v.sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH -d $O/$v/sq3 --output-format csv -- python3 $R/tools/env_bench.py > $O/$v.sq3.log 2>&1 || echo "pass 3 of $v failed"
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'bwd_pmc')
out = open(os.path.join(O, 'summary.txt'), 'w')
for v in 'AB':
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f'{O}/{v}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'env_bwd' in row['Kernel_Name']:
                acc[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    print(f'== build {v}: env_bwd_kernel, per launch', file=out)
    for k in sorted(acc):
        print(f'  {k:26s} {acc[k]/n[k]:16.0f}   ({n[k]} launches)', file=out)
print(open(os.path.join(O, 'summary.txt')).read())
PY
