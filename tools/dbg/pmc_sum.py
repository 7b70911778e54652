import csv, glob, collections, sys
tag = sys.argv[1]; mbs = float(sys.argv[2]) if len(sys.argv) > 2 else 0
for d in ('a', 'b', 'c'):
    for f in glob.glob(f'gpurun_out/pmc_{tag}_{d}/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in agg.items():
            if k.startswith('k_analyse') or k.startswith('k_search') or k.startswith('k_rca'):
                print(k, {a: (f"{b:.3g}" + (f" ({b/mbs:.0f}/MB)" if mbs else "")) for a, b in v.items()})
