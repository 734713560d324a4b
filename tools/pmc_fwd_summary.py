"""Summary of tools/pmc_fwd_libs.sh: per run directory the counters of chain_fwd_kernel, averaged per launch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, '*.[ab]'))):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    acc, launches = collections.defaultdict(float), collections.defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f)):
            if 'chain_fwd_kernel' not in row.get('Kernel_Name', ''):
                continue
            acc[row['Counter_Name']] += float(row['Counter_Value'])
            launches[row['Counter_Name']].add(row.get('Dispatch_Id'))
    if not acc:
        print(os.path.basename(d), 'no chain_fwd_kernel rows'); continue
    print(os.path.basename(d) + ': ' + '  '.join(f'{k} {acc[k]/max(1, len(launches[k]))/1e6:.1f}M' for k in sorted(acc)))
