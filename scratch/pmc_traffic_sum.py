import sys, glob, csv, collections, re
tag, entries = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(tag + '/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        name = re.sub(r'\(anonymous namespace\)::', '', k)
        name = re.sub(r'^void ', '', name).split('(')[0]
        if not name.startswith('k_'): continue
        agg[name][row['Counter_Name']] += float(row['Counter_Value'])
GiB = 1 << 30
# calibration: k_mask_rows<4> moved exactly 1 GiB in and 1 GiB out with 16-B lanes; <1> the same +0.4 % with 4-B lanes
cal = {}
for v, nbytes in (('k_mask_rows<4>', (1 << 20) * 256 * 4), ('k_mask_rows<1>', (1 << 20) * 257 * 4)):
    if v in agg:
        cal[v] = (nbytes / (agg[v]['FETCH_SIZE'] * 1024), nbytes / (agg[v]['WRITE_SIZE'] * 1024))
print('# FETCH_SIZE / WRITE_SIZE are in KiB; calibration factors = true bytes / reported bytes (read, write)')
for k, (fr, fw) in cal.items():
    print(f'# {k}: read x{fr:.3f}  write x{fw:.3f}')
f1 = cal.get('k_mask_rows<1>', (1.0, 1.0))
print(f"{'kernel':28s} {'FETCH KiB':>14s} {'WRITE KiB':>14s} {'read GB (x%.2f)' % f1[0]:>16s} {'write GB (x%.2f)' % f1[1]:>17s} {'B/entry':>9s}")
tot = collections.defaultdict(float)
for name, c in sorted(agg.items()):
    rd, wr = c['FETCH_SIZE'] * 1024 * f1[0], c['WRITE_SIZE'] * 1024 * f1[1]
    print(f"{name:28s} {c['FETCH_SIZE']:14.0f} {c['WRITE_SIZE']:14.0f} {rd / 1e9:16.3f} {wr / 1e9:17.3f} {(rd + wr) / entries:9.0f}")
    grp = 'forward' if ('fwd' in name or name == 'k_graph_reduce') else ('backward' if 'bwd' in name else None)
    if grp: tot[grp] += rd + wr
for g, b in tot.items():
    print(f'# hotk_attn22_{g}: {b / 1e9:.2f} GB HBM traffic per launch = {b / entries:.0f} B/entry (k_graph_reduce counted once in forward; it also runs once in backward)')
