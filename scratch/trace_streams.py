"""Per-queue busy time from a rocprofv3 kernel trace (scratch tool): python scratch/trace_streams.py <kernel_trace.csv> [skip_frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[int(n * 0.5):int(n * 0.9)]                     # the timed region's middle
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
qkey = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
import re
def short(n):
    m = re.findall(r"(\w+Functor\w*<[\w:]+>|\w+_kernel_cuda|\w+_kernel_impl|k_\w+|Cijk_\w{10}.{0,40}MT\w+|nccl\w+|index\w+|\w+_kernel\b)", n)
    keep = [x for x in m if not x.startswith(("elementwise_kernel", "vectorized_elementwise", "gpu_kernel"))]
    return (" ".join(dict.fromkeys(keep[:3])) or n)[:90]
calls = collections.defaultdict(lambda: collections.defaultdict(int))
busy = collections.defaultdict(int); cnt = collections.defaultdict(int); names = collections.defaultdict(lambda: collections.defaultdict(int))
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    q = r[qkey]; busy[q] += d; cnt[q] += 1; names[q][short(r["Kernel_Name"])] += d; calls[q][short(r["Kernel_Name"])] += 1
print("window %.3f ms, %d kernels, key %s" % ((t1 - t0) / 1e6, len(rows), qkey))
for q in busy:
    print("queue %s: busy %.3f ms (%.1f%%), %d launches" % (q, busy[q] / 1e6, 100.0 * busy[q] / (t1 - t0), cnt[q]))
    for k, v in sorted(names[q].items(), key=lambda kv: -kv[1])[:45]:
        print("    %6.1f%%  %5d  %s" % (100.0 * v / busy[q], calls[q][k], k))

if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    # one window of ~3 ms from the middle: every kernel with its start offset, duration and stream
    mid = rows[len(rows) // 2]
    w0 = int(mid["Start_Timestamp"])
    print("\ntimeline (us from window start; stream; duration; kernel)")
    for r in rows:
        a = int(r["Start_Timestamp"]) - w0
        if 0 <= a < 4_000_000:
            print("%9.1f  q%-2s %7.1f  %s" % (a / 1e3, r[qkey], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short(r["Kernel_Name"])[:60]))
