import sys, os, torch
sys.path.insert(0, '/root/repo')
import bliss_gnn_amd as bg
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader
dev = torch.device('cuda:0')
cfg = CONFIGS['cora']
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
feats, labels, train_nid = node_data(cfg["num_nodes"], cfg["feat"], cfg["classes"], cfg["n_train"], seed=1, device=dev)
g = bg.Graph(ip, ix, ei); g.edata["w"] = bg.normalized_edata(g)
s = bg.PoissonBanditLadiesSampler(cfg["fanouts"], eta=0.1)
loader = BatchLoader(train_nid, cfg["batch"], seed=2).forever()
torch.manual_seed(3)
for i in range(6):
    try:
        inp, _, blocks = s.sample_blocks(g, next(loader))
        print(i, "ok", [(b._counts.S, b._counts.E, b._counts.C, b._counts.K, b._counts.B, b._counts.err) for b in blocks])
    except RuntimeError as e:
        print(i, "ERR", e)
        raw = s._engine.counts_host.numpy().reshape(-1, 10)
        print(raw)
        break
