"""Per-kernel averages of arbitrary rocprofv3 --pmc counters.  usage: pmc_kernels.py <dir> [name filter]"""
import collections, csv, glob, re, sys
tot, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'^void ', '', k).split('(')[0]
        if not k.startswith('k_'):
            continue
        key = (k, r.get('Grid_Size', ''))
        tot[key][r['Counter_Name']] += float(r['Counter_Value']); n[key][r['Counter_Name']] += 1
names = sorted({c for v in tot.values() for c in v})
print('kernel grid ' + ' '.join(names))
for key in sorted(tot):
    print(key[0], key[1], ' '.join('%.3g' % (tot[key][c] / max(n[key][c], 1)) for c in names))
