"""Turn the rocprofv3 CSVs of one round (kernel stats + PMC passes collected on the GPU
box under gpurun_out/) into the small summaries committed under profiles/.

usage: python tools/summarize_profiles.py gpurun_out/prof_r01 profiles/r01
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or '.', exist_ok=True)


def short(name):
    """Kernel names as bench.py prints them."""
    m = re.search(r'sos_ckpt_kernel<(\d), (\d)(?:, \w+)?>', name)
    if m:
        return ('sos_ckpt<S=%s+%s,filt+env_state>' if m.group(1) != '0' else 'sos_ckpt<S=%s+%s,env_state>') % m.groups()
    m = re.search(r'chain_fwd_kernel<(\d), (\d), \d+(?:, \w+)*>', name)
    if m:
        return 'chain_fwd<S=%s+%s,filt+env_state+psd>' % m.groups()
    m = re.search(r'env_bwd_kernel<(\d)(?:, \w+)*>', name)
    if m:
        return 'env_bwd<S=%s>' % m.group(1)
    m = re.search(r'sos_scan_kernel<(\d)>', name)
    if m:
        return 'sos_scan<S=%s,filt>' % m.group(1)
    for key, tag in (('spec_fast_kernel', 'spectrogram'), ('spec2_kernel', 'spectrogram'),
                     ('spec_generic', 'spectrogram_generic'), ('synth', 'synth')):
        if key in name:
            return tag
    return name[:60]


stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))
if stats:
    shutil.copy(stats[0], dst + '_kernel_stats.csv')
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('fetch', 'write', 'sq', 'sq2'):
    for f in glob.glob(os.path.join(src, d, '*', '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            counters[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in counters.items():
    if k.startswith('__amd') or k == 'synth':
        continue
    row = {c: (sum(v)/len(v)) for c, v in cs.items()}
    if 'FETCH_SIZE' in row and 'WRITE_SIZE' in row:
        # MI355X_MICROARCH.md / HBM: FETCH_SIZE and WRITE_SIZE are in KiB-like units of
        # 1024 B; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream
        row['hbm_read_bytes'] = 2.0*row['FETCH_SIZE']*1024
        row['hbm_write_bytes'] = row['WRITE_SIZE']*1024
        row['hbm_bytes'] = row['hbm_read_bytes'] + row['hbm_write_bytes']
    out[k] = row
json.dump(out, open(dst + '_pmc.json', 'w'), indent=1, sort_keys=True)
if len(sys.argv) > 3:
    # shape of the profiled run "C,T,nfft,hop" -> the file bench.py reads `roofline.traffic` from
    shape = [int(v) for v in sys.argv[3].split(',')]
    json.dump({'shape': shape, 'source': os.path.basename(dst) + '_pmc.json',
               'kernels': {k: {'hbm_bytes': row['hbm_bytes'], 'hbm_read_bytes': row['hbm_read_bytes'],
                               'hbm_write_bytes': row['hbm_write_bytes']}
                           for k, row in out.items() if 'hbm_bytes' in row}},
              open(os.path.join(os.path.dirname(dst) or '.', 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
for k, row in out.items():
    if 'hbm_bytes' in row:
        print(f"{k:26s} read {row['hbm_read_bytes']/1e9:7.2f} GB  write {row['hbm_write_bytes']/1e9:7.2f} GB")
