import sys; sys.path.insert(0,'.')
import torch
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import chung_lu_csc
from bliss_gnn_amd.train import BatchLoader, GraphedTrainStep
def log(*a):
    print(*a, flush=True)
cuda=torch.device('cuda:0')
ip, ix, ei = chung_lu_csc(8000, 160000, seed=12)
feats = torch.randn(8000, 64, generator=torch.Generator().manual_seed(2)).bfloat16()
labels = torch.randint(0, 5, (8000,), generator=torch.Generator().manual_seed(3))
fan, bs = [400, 200, 100], 64
ids = torch.arange(8000, dtype=torch.int32, device=cuda)
g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
g.edata["w"] = bg.normalized_edata(g)
sampler = bg.PoissonBanditLadiesSampler(fan, eta=0.1)
torch.manual_seed(0)
model = SAGE(64, 32, 5, 3, torch.relu, 0.0).to(cuda).bfloat16()
gs = GraphedTrainStep(g, sampler, model, bs)
l2 = BatchLoader(ids, bs, seed=5).forever()
torch.manual_seed(9)
gs.calibrate(l2, steps=3)
log('caps', sampler._engine.caps)
eng=sampler._engine
mode = sys.argv[1]
if mode == 'eager':
    for i in range(12):
        gs.seeds.copy_(next(l2)); eng.stage_rng_from_torch()
        loss = gs._body(); gs._finish()
        log('eager static step', i, float(loss), [(c.S,c.E,c.C,c.K,c.B,c.err) for c in gs.last_counts])
else:
    gs.capture(l2, warmup=3)
    log('captured', [(c.S,c.E,c.C,c.K,c.B,c.err) for c in gs.last_counts])
    for i in range(8):
        gs(next(l2)); log('replay', i, float(gs.loss), [(c.S,c.E,c.C,c.K,c.B,c.err) for c in gs.last_counts])
log('done')
