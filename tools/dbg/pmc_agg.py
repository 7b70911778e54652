"""Sum the counters of a rocprofv3 --pmc output directory per kernel (the raw CSV has a row per dispatch and counter and does not
fit gpurun's 64 MiB return limit at thousands of GOPs):  python tools/dbg/pmc_agg.py <dir> <out.json>  (then delete <dir>)"""
import collections, csv, glob, json, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
json.dump({k: {c: {"sum": v, "dispatches": n[k][c]} for c, v in d.items()} for k, d in tot.items() if k.startswith('k_')}, open(sys.argv[2], 'w'), indent=1)
