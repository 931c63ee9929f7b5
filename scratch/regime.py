"""Fast vs slow stretches of the free-running loop: per-kernel mean duration in the steps of either kind.
usage: regime.py <rocprof dir>"""
import csv, glob, re, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
st = np.array([int(r['Start_Timestamp']) for r in rows]); en = np.array([int(r['End_Timestamp']) for r in rows])
idx = np.array([i for i, r in enumerate(rows) if 'k_exp3_update_multi' in r['Kernel_Name']])
span = np.diff(st[idx]) / 1e3
print('steps', len(span), 'median span', np.median(span))
# the free-running stretch: the last 700 steps
lo = max(0, len(span) - 700)
sp = span[lo:]
# a step is "slow" when an F.normalize pass that really streams a row starts inside it
nrm = [i for i, r in enumerate(rows) if 'k_normalize_rows' in r['Kernel_Name'] and (en[i] - st[i]) > 20000]
slow = np.zeros(len(sp), bool)
for i in nrm:
    k = np.searchsorted(st[idx], st[i]) - 1 - lo
    if 0 <= k < len(sp): slow[k] = True
print('passes', len(nrm), 'mean us', np.mean([(en[i] - st[i]) / 1e3 for i in nrm]) if nrm else 0)
runs = []; cur = slow[0]; n = 0
for s in slow:
    if s == cur: n += 1
    else: runs.append((int(cur), n)); cur = s; n = 1
runs.append((int(cur), n)); print('runs (slow?, steps):', [r for r in runs if r[1] > 4])
print('fast mean', sp[~slow].mean(), 'slow mean', sp[slow].mean())
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:60]
acc = {0: collections.defaultdict(list), 1: collections.defaultdict(list)}
for k in range(len(sp)):
    a, b = idx[lo + k], idx[lo + k + 1]
    seen = collections.Counter()
    for i in range(a, b):
        nm = short(rows[i]['Kernel_Name']); seen[nm] += 1
        acc[int(slow[k])][(nm, seen[nm])].append((en[i] - st[i]) / 1e3)
keys = sorted(acc[0].keys() & acc[1].keys(), key=lambda k: -(np.mean(acc[1][k]) - np.mean(acc[0][k])))
tot = 0
for k in keys[:25]:
    d = np.mean(acc[1][k]) - np.mean(acc[0][k]); tot += d
    print(f"{k[0]:60s} #{k[1]} fast {np.mean(acc[0][k]):8.2f} slow {np.mean(acc[1][k]):8.2f} diff {d:7.2f}")
print('sum of all diffs', sum(np.mean(acc[1][k]) - np.mean(acc[0][k]) for k in keys))

# one slow step, kernel by kernel
ks = np.where(slow)[0]
if len(ks):
    k = ks[len(ks) // 2]
    a, b = idx[lo + k], idx[lo + k + 1]
    t0 = st[a]
    qs = sorted({int(rows[i]['Queue_Id']) for i in range(a, b)})
    print('--- slow step', k, 'span', sp[k])
    for i in range(a, b):
        q = qs.index(int(rows[i]['Queue_Id']))
        print(f"{(st[i]-t0)/1e3:8.1f} {(en[i]-st[i])/1e3:7.1f} {'          '*q}q{q} {short(rows[i]['Kernel_Name'])[:50]}")
