"""Eager sample_blocks on the Reddit-like graph; prints the library profiler's per-kernel averages (all layers pooled)."""
import sys, torch
sys.path.insert(0, '.')
import bliss_gnn_amd as bg
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader
from bliss_gnn_amd import roofline
dev = torch.device('cuda:0')
cfg = CONFIGS['reddit']
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
feats, labels, train_nid = node_data(cfg["num_nodes"], cfg["feat"], cfg["classes"], cfg["n_train"], seed=1, device=dev)
g = bg.Graph(ip, ix, ei); g.edata["w"] = bg.normalized_edata(g)
s = bg.PoissonBanditLadiesSampler(cfg["fanouts"], eta=0.1)
loader = BatchLoader(train_nid, cfg["batch"], seed=2).forever()
def once():
    try:
        s.sample_blocks(g, next(loader))
    except RuntimeError:
        torch.cuda.synchronize()
for _ in range(5): once()
t = roofline.KernelTimer(); t.enable("all")
for _ in range(20): once()
torch.cuda.synchronize()
r = t.read()
print({k: round(v["avg_us"], 1) for k, v in r.items() if k in ("k_col_sums", "k_bin_scatter", "k_bin_reduce", "k_block_pass1", "k_block_pass2", "k_seg_scan", "k_cand_number")})
