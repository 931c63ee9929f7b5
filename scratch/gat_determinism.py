"""Is the small GATv2 train step (tests/test_gpu_parity.py::test_gatv2_model_train_step) deterministic run to run?  usage: python scratch/gat_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import GATv2
from bliss_gnn_amd.synth import chung_lu_csc
cuda = torch.device("cuda", 0)
ip, ix, ei = chung_lu_csc(4000, 60000, seed=45)
feats = torch.randn(4000, 48, generator=torch.Generator().manual_seed(1)).bfloat16()
labels = torch.randint(0, 6, (4000,), generator=torch.Generator().manual_seed(2))
runs = []
for rep in range(4):
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
    g.edata["w"] = bg.normalized_edata(g)
    s = bg.PoissonBanditLadiesSampler([300, 150, 80], eta=0.1, model="gat")
    torch.manual_seed(0)
    model = GATv2(3, 48, 16, 6, [4, 4, 1], torch.nn.functional.elu, 0.1, 0.1, 0.2, True).to(cuda).bfloat16()
    opt = torch.optim.Adam(model.parameters(), lr=0.002)
    losses, grads = [], []
    for step in range(4):
        torch.manual_seed(step)
        inp, outp, blocks = s.sample_blocks(g, torch.arange(64, dtype=torch.int32, device=cuda))
        pred = model(blocks, blocks[0].srcdata["features"])
        loss = torch.nn.functional.cross_entropy(pred, blocks[-1].dstdata["labels"])
        opt.zero_grad(); loss.backward(); opt.step()
        s.exp3(blocks, g)
        losses.append(float(loss))
        grads.append(float(sum(p.grad.float().abs().sum() for p in model.parameters() if p.grad is not None)))
    runs.append((losses, grads, [p.detach().float().cpu().clone() for p in model.parameters()], s.exp3_weights.cpu().view(torch.int16).clone()))
    print(rep, losses, [round(x, 4) for x in grads])
for r in runs[1:]:
    print("params equal", all(torch.equal(a, b) for a, b in zip(runs[0][2], r[2])), "exp3 equal", torch.equal(runs[0][3], r[3]))
