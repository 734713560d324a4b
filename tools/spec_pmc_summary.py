"""Summary of tools/spec_pmc.sh: per window length the HBM bytes of one hipdsp_spectrogram call (all its kernels; FETCH_SIZE x 2
on gfx950 -- MI355X_MICROARCH.md -- + WRITE_SIZE, both in units of 1024 bytes) against the algorithmic bytes
(4 per sample read + 4 per bin written, 8 with the dB image)."""
import collections, csv, glob, os, sys
root, sizes = sys.argv[1], [int(a) for a in sys.argv[2:]]
C, T = 64, 120*96000
DB_ARG = {'spec_chipx_kernel': 0, 'spec_chip_kernel': 0, 'spec_pack_kernel': -1, 'spec_wgs_kernel': -2, 'spec_fast_kernel': -2, 'spec_wg_kernel': -1}


def db_of(name):
    """True / False for kernels templated on the dB output, None for the others (they run for both variants alike)."""
    for key, pos in DB_ARG.items():
        if key in name and '<' in name:
            return name[name.index('<') + 1:name.index('>')].split(', ')[pos] == 'true'
    return None


for n in sizes:
    hop = n//2
    t = T if n >= 64 else T//8
    nd = (t + hop - 1)//hop
    F = n//2 + 1
    per = {}
    for what in ('fetch', 'write'):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(root, f'{what}_{n}', '*', '*_counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if 'synth' in r['Kernel_Name'] or r['Kernel_Name'].startswith('__amd'):
                    continue
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
        per[what] = acc
    for db in (False, True):
        rd = wr = 0.0
        for k in sorted(set(per['fetch']) & set(per['write'])):
            kdb = db_of(k)
            if kdb is not None and kdb != db:
                continue
            # spec_sizes_bench.py runs 6 calls per variant; a kernel that is launched several times per call (the four-step
            # path's batches) counts with all its launches
            share = 6.0 if kdb is not None else 12.0
            rd += 2048.0*sum(per['fetch'][k])/share
            wr += 1024.0*sum(per['write'][k])/share
        alg = 4.0*C*t + (8.0 if db else 4.0)*C*nd*F
        if rd + wr > 0:
            print(f'nfft {n:6d} {"PSD+dB" if db else "PSD   "}: HBM read {rd/1e9:7.3f} GB  write {wr/1e9:7.3f} GB  '
                  f'algorithmic {alg/1e9:7.3f} GB  ratio {(rd + wr)/alg:5.3f}')
