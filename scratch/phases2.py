"""Phase boundaries of the pipelined loop in free-running mode (no sync between pairs): timing events on the main stream
between the graph replays of PipelinedTrainStep._replay / _replay_sampler."""
import sys, time, torch
sys.path.insert(0, '.')
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader, PipelinedTrainStep
dev = torch.device('cuda:0')
cfg = CONFIGS['reddit']
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
feats, labels, train_nid = node_data(cfg["num_nodes"], cfg["feat"], cfg["classes"], cfg["n_train"], seed=1, device=dev)
g = bg.Graph(ip, ix, ei, ndata={"features": feats, "labels": labels}); g.edata["w"] = bg.normalized_edata(g)
sampler = bg.PoissonBanditLadiesSampler(cfg["fanouts"], eta=0.1)
torch.manual_seed(1234)
model = SAGE(cfg["feat"], 256, cfg["classes"], 3, torch.relu, 0.1).to(dev).bfloat16()
loader = BatchLoader(train_nid, cfg["batch"], seed=2).forever()
step = PipelinedTrainStep(g, sampler, model, cfg["batch"])
step.calibrate(loader, steps=8); step.capture(loader, warmup=2, tune_gemm=True)
step.run(loader, 50)
torch.cuda.synchronize()
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); marks.append((name, e))


class W:
    def __init__(s, gr, a, b): s.gr, s.a, s.b = gr, a, b

    def replay(s):
        if s.a: mark(s.a)
        s.gr.replay()
        if s.b: mark(s.b)


bmarks = []


class WB:
    def __init__(s, gr): s.gr = gr

    def replay(s):
        a = torch.cuda.Event(enable_timing=True); a.record(torch.cuda.current_stream())
        s.gr.replay()
        b = torch.cuda.Event(enable_timing=True); b.record(torch.cuda.current_stream())
        bmarks.append((a, b))


if step.use_flags:
    step.g_main = [W(x, "F_start", "S_end") for x in step.g_main]
    step.g_bwd = [WB(x) for x in step.g_bwd]
else:
    step.g_fwd = [W(x, "F_start", "X_end") for x in step.g_fwd]
    step.g_smp = [W(x, None, "S_end") for x in step.g_smp]
eng = sampler._engine
orig_end = eng.static_rng_end


def end(slot):
    orig_end(slot); mark("tail_end")


eng.static_rng_end = end
N = 40
step.run(loader, N)
torch.cuda.synchronize()
acc, cnt = {}, {}
for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
    k = n0 + " -> " + n1
    acc[k] = acc.get(k, 0.0) + e0.elapsed_time(e1) * 1e3; cnt[k] = cnt.get(k, 0) + 1
tot = 0
for k in acc:
    print("%8.1f us  x%d  %s" % (acc[k] / cnt[k], cnt[k], k)); tot += acc[k] / cnt[k]
print("sum", tot)

# backward stream against the critical stream: does B (output layer, loss, backward, Adam) end before the sampler does?
import numpy as np
s_end = [e for n, e in marks if n == "S_end"]
f_start = [e for n, e in marks if n == "F_start"]
if bmarks and len(s_end) == len(bmarks):
    lag = np.array([se.elapsed_time(be) * 1e3 for se, (bs, be) in zip(s_end, bmarks)])       # > 0: B ends after S
    dur = np.array([fs.elapsed_time(be) * 1e3 for fs, (bs, be) in zip(f_start, bmarks)])     # F_start -> B_end
    print("B_end - S_end (us): median %.1f p10 %.1f p90 %.1f ; share of halves with B later than S: %.2f" % (
        np.median(lag), np.percentile(lag, 10), np.percentile(lag, 90), (lag > 0).mean()))
    print("F_start -> B_end (us): median %.1f" % np.median(dur))
    # the graph boundary itself: S_end -> next F_start in the halves whose B had finished well before S
    gaps = np.array([se.elapsed_time(fs) * 1e3 for se, fs in zip(s_end[:-1], f_start[1:])])
    l = lag[:len(gaps)]
    for name, m in (("B earlier than S by > 20 us", l < -20), ("B within 20 us of S", np.abs(l) <= 20), ("B later by > 20 us", l > 20)):
        if m.any():
            print("%-32s n=%3d  gap median %.1f us (min %.1f)" % (name, m.sum(), np.median(gaps[m]), gaps[m].min()))
