"""How much of a bandit row do the updates ever touch?  Distinct 32-element (64-byte) lines written by exp3 over N steps."""
import sys, torch
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
step.calibrate(loader, steps=8); step.capture(loader, warmup=2)
E = g.num_edges(); nl = (E + 31) // 32
w0 = sampler._w_pos.clone()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for chunk in range(N // 200):
    step.run(loader, 100)
    torch.cuda.synchronize()
    # a touched element differs from what an untouched one became: within a row all untouched elements are equal
    for l in range(3):
        row = sampler._w_pos[l].view(torch.int16)
        base = torch.mode(row[::9973]).values          # the value the untouched majority holds
        ch = (row != base)
        lines = torch.zeros(nl * 32, dtype=torch.bool, device=dev); lines[:E] = ch
        tl = lines.view(nl, 32).any(1).sum().item()
        print("steps %4d row %d: touched elements %9d (%.2f %%), touched 64-byte lines %8d (%.2f %% of %d)" % (
            (chunk + 1) * 200, l, int(ch.sum()), 100.0 * float(ch.sum()) / E, tl, 100.0 * tl / nl, nl))
