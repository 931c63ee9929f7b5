import sys; sys.path.insert(0,'.')
import torch, gc
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import chung_lu_csc
from bliss_gnn_amd.train import BatchLoader, GraphedTrainStep
def log(*a): print(*a, flush=True)
level=int(sys.argv[1])
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
eng=sampler._engine
def body():
    inp,outp,mfgs = sampler.sample_blocks_static(g, gs.seeds)
    if level==1: return None
    x = mfgs[0].srcdata["features"]; y = mfgs[-1].dstdata["labels"]
    pred = model(mfgs, x); loss = gs.loss_fn(pred, y)
    if level==2: return loss.detach()
    gs.opt.zero_grad(set_to_none=True); loss.backward()
    if level==3: return loss.detach()
    gs.opt.step()
    if level==4: return loss.detach()
    sampler.exp3(mfgs, g)
    return loss.detach()
gs._body = body
gs.capture(l2, warmup=3)
log('level',level,'captured', [(c.S,c.E,c.C,c.K,c.B,c.err) for c in gs.last_counts])
for i in range(6):
    gs(next(l2)); log('level',level,'replay', i, None if gs.loss is None else float(gs.loss), [(c.K,c.B,c.err) for c in gs.last_counts])
log('level',level,'done')
