"""Reduce a rocprofv3 --pmc counter_collection.csv to per-kernel means: python tools/pmc_kernel.py <dir> [name filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        if flt in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, 'launches', max(len(v) for v in cs.values()))
