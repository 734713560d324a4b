"""Per process of tools/bwd_tlb_probe.sh: mean duration and translation counters of env_bwd_kernel launches."""
import csv, glob, os, sys, collections
for run in sorted(glob.glob(os.path.join(sys.argv[1], 'run*'))):
    if not os.path.isdir(run):
        continue
    cc = glob.glob(os.path.join(run, '*', '*_counter_collection.csv'))
    kt = glob.glob(os.path.join(run, '*', '*_kernel_trace.csv'))
    if not cc or not kt:
        continue
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(kt[0]))
           if 'env_bwd' in r['Kernel_Name']]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if 'env_bwd' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    line = f'{os.path.basename(run)}: env_bwd {len(dur)} launches, median {sorted(dur)[len(dur)//2]:.3f} ms'
    for k, v in sorted(agg.items()):
        line += f' | {k} {sum(v)/len(v)/1e6:.2f} M'
    print(line)
