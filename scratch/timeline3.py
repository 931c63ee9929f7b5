"""One steady-state train step of the pipelined loop, kernel by kernel with queue ids: the window between two consecutive
feature gathers (k_gather_rows_norm opens the forward pass).  usage: timeline3.py <rocprof dir> [which]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_exp3_update_multi' in r['Kernel_Name']]     # once per step (X follows F)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) * 2 // 3
t0, t1 = int(rows[idx[which]]['Start_Timestamp']), int(rows[idx[which + 1]]['Start_Timestamp'])
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:56]
qs = sorted({int(r['Queue_Id']) for r in step})
for r in step:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    q = qs.index(int(r['Queue_Id']))
    print(f"{s:8.1f} {d:7.1f} {'          ' * q}q{q} {short(r['Kernel_Name'])}")
print('step span us', (t1 - t0) / 1e3, 'n kernels', len(step))
