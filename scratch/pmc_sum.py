import sys, glob, csv, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(tag + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[1] if row['Kernel_Name'].startswith('(anonymous') else row['Kernel_Name']
        k = row['Kernel_Name']
        if 'k_' not in k: continue
        name = k.split('::')[-1].split('(')[0]
        agg[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name, cs in agg.items():
    print(name)
    for c, vals in sorted(cs.items()):
        print(f'   {c:28s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})')
