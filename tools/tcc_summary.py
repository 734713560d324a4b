"""Counters of tools/tcc_probe.sh per kernel, normalised per launch and per GB moved where that helps."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for kind in ('bwd', 'copy'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for run in sorted(glob.glob(os.path.join(root, kind + '[0-9]'))):
        for f in glob.glob(os.path.join(run, '*', '*_counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for f in glob.glob(os.path.join(run, '*', '*_kernel_trace.csv')):
            for r in csv.DictReader(open(f)):
                dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e6)
    for k, v in agg.items():
        if not ('env_bwd' in k or 'copy' in k.lower()):
            continue
        d = sorted(dur[k])
        print(f'== {k}: {len(d)} launches, median {d[len(d)//2]:.3f} ms')
        for c, x in sorted(v.items()):
            print(f'     {c:42s} {sum(x)/len(x)/1e6:12.3f} M per launch')
