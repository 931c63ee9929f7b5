"""Where a pipelined pair spends its time, measured with events on both streams (no profiler in the way)."""
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
step.run(loader, 20)
eng = sampler._engine
main, side = torch.cuda.current_stream(), step.side
E = lambda: torch.cuda.Event(enable_timing=True)
acc = {}
N = 30
for it in range(N):
    step.seeds2[1].copy_(next(loader)); step.seeds2[0].copy_(next(loader))
    ev = {}
    side.wait_stream(main)
    t0 = E(); t0.record(main)
    for h, (cur, nxt, chain) in enumerate(((0, 1, it > 0), (1, 0, True))):
        eng.static_rng_begin(chain)
        with torch.cuda.stream(side):
            a = E(); a.record(side); step.g_fwd[cur].replay(); b = E(); b.record(side)
        main.wait_stream(side)
        c = E(); c.record(main); step._replay_sampler(nxt); eng.static_rng_end(nxt); d = E(); d.record(main)
        with torch.cuda.stream(side):
            e = E(); e.record(side); step.g_bwd[cur].replay(); f = E(); f.record(side); side.wait_stream(main)
        ev[h] = (a, b, c, d, e, f)
    main.wait_stream(side)
    t1 = E(); t1.record(main)
    torch.cuda.synchronize()
    for h in (0, 1):
        a, b, c, d, e, f = ev[h]
        for k, v in (("F+X", a.elapsed_time(b)), ("X_end->S_start", b.elapsed_time(c)), ("S", c.elapsed_time(d)), ("X_end->B_start", b.elapsed_time(e)),
                     ("B", e.elapsed_time(f)), ("half_start->F_start", (t0 if h == 0 else ev[0][3]).elapsed_time(a))):
            acc[k] = acc.get(k, 0.0) + v / (2 * N)
    acc["pair"] = acc.get("pair", 0.0) + t0.elapsed_time(t1) / N
print({k: round(1e3 * v, 1) for k, v in acc.items()}, "(us; this loop syncs per pair)")
