"""Per-queue view of one pipelined pair: prints kernels grouped by phase with queue ids.  usage: timeline2.py <dir> [k]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rng_ctl_init' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
which -= which % 2
step = rows[idx[which]:idx[which + 2]]
t0 = int(step[0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:48]
for r in step:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    q = int(r['Queue_Id'])
    print(f"{s:8.1f} {d:7.1f} {'      ' * (q % 5)}q{q} {short(r['Kernel_Name'])}")
print('pair span us', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'n kernels', len(step))
