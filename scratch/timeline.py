"""One step's kernels in time order from a rocprofv3 --kernel-trace run.  usage: timeline.py <rocprof out dir> [step index]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rng_ctl_init' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
step = rows[idx[which]:idx[which + 1]]
t0 = int(step[0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:70]
prev_end = t0
for r in step:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev_end) / 1e3
    prev_end = max(prev_end, int(r['End_Timestamp']))
    print(f"{s:8.1f} {d:7.1f} gap={gap:6.1f} q={r['Queue_Id']} {short(r['Kernel_Name'])}")
print('step span us', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'n kernels', len(step), 'steps in trace', len(idx))
