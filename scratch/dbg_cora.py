"""Where does the Cora-like run first see a non-finite value?  Eager TrainStep, step by step."""
import sys, torch
sys.path.insert(0, '.')
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader, TrainStep
dev = torch.device('cuda:0')
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cora']
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
feats, labels, train_nid = node_data(cfg["num_nodes"], cfg["feat"], cfg["classes"], cfg["n_train"], seed=1, device=dev)
g = bg.Graph(ip, ix, ei, ndata={"features": feats, "labels": labels}); g.edata["w"] = bg.normalized_edata(g)
sampler = bg.PoissonBanditLadiesSampler(cfg["fanouts"], eta=0.1)
torch.manual_seed(1234)
model = SAGE(cfg["feat"], 256, cfg["classes"], 3, torch.relu, 0.1).to(dev).bfloat16()
loader = BatchLoader(train_nid, cfg["batch"], seed=2).forever()
torch.manual_seed(3)
step = TrainStep(g, sampler, model, lr=0.002)
for i in range(200):
    try:
        loss = step(next(loader))
    except RuntimeError as e:
        print(i, "ERR", str(e)[:200]); break
    w = sampler.exp3_weights.float()
    mfgs = step.last["mfgs"]
    en = [float(m.srcdata["embed_norm"].float().max()) for m in mfgs]
    fin = bool(torch.isfinite(w).all())
    if i % 10 == 0 or not fin:
        print(i, "loss %.4f" % float(loss), "w finite", fin, "w max %.3e min %.3e" % (float(w.max()), float(w.min())), "embed_norm max", en,
              "rowsum", [float(r.sum()) for r in w])
    if not fin:
        break
