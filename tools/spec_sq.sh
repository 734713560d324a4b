#!/bin/bash
# SQ counters of hipdsp_spectrogram per window length (instruction mix and wait cycles of the PSD kernels; the passes of
# tools/profile_round.sh, on tools/spec_sizes_bench.py):  gpurun -- 'bash tools/spec_sq.sh 2048 8192'  ->  gpurun_out/spec_sq/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/spec_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export WARM_CALLS=0 TIMED_CALLS=4      # the summaries count on 2 + 4 calls per variant and no other kernel
for n in "$@"; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/sq_$n --output-format csv -- python3 $R/tools/spec_sizes_bench.py $n > $O/sq_$n.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq2_$n --output-format csv -- python3 $R/tools/spec_sizes_bench.py $n > $O/sq2_$n.log 2>&1 || exit 1
  echo "nfft $n done"
done
python3 - "$O" "$@" <<'PY' | tee $O/summary.txt
import collections, csv, glob, os, sys
root, sizes = sys.argv[1], [int(a) for a in sys.argv[2:]]
C, T = 64, 120*96000
for n in sizes:
    frames = C*((T + n//2 - 1)//(n//2))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ('sq', 'sq2'):
        for f in glob.glob(os.path.join(root, f'{d}_{n}', '*', '*_counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                if 'synth' in k or k.startswith('__amd'):
                    continue
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        row = {c: sum(v)/len(v) for c, v in cs.items()}
        print(f'nfft {n}: {k[:110]}')
        print('    per frame: ' + ', '.join(f'{c[3:]} {row[c]/frames:.0f}' for c in sorted(row) if c.startswith('SQ_INSTS')))
        wc = row.get('SQ_WAVE_CYCLES', 0)
        if wc:
            print(f"    waves {row['SQ_WAVES']:.0f}, wave cycles per frame {wc/frames:.0f}, waiting for any instruction {100*row['SQ_WAIT_INST_ANY']/wc:.0f} %, "
                  f"VALU active / busy cycles {row.get('SQ_ACTIVE_INST_VALU', 0)/row['SQ_BUSY_CYCLES']:.2f}, LDS active / busy {row.get('SQ_ACTIVE_INST_LDS', 0)/row['SQ_BUSY_CYCLES']:.2f}, "
                  f"bank conflict cycles / LDS active {row.get('SQ_LDS_BANK_CONFLICT', 0)/max(row.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f}")
PY
