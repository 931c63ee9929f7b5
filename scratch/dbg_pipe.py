"""Bisect the capture_end crash of PipelinedTrainStep.  usage: dbg_pipe.py <variant>"""
import sys, torch
sys.path.insert(0, '.')
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import chung_lu_csc
from bliss_gnn_amd.train import BatchLoader, PipelinedTrainStep
variant = sys.argv[1]
cuda = torch.device('cuda:0')
ip, ix, ei = chung_lu_csc(8000, 160000, seed=12)
feats = torch.randn(8000, 64).bfloat16(); labels = torch.randint(0, 5, (8000,))
ids = torch.arange(8000, dtype=torch.int32, device=cuda)
g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
g.edata["w"] = bg.normalized_edata(g)
sampler = bg.PoissonBanditLadiesSampler([400, 200, 100], eta=0.1)
model = SAGE(64, 32, 5, 3, torch.relu, 0.0).to(cuda).bfloat16()
step = PipelinedTrainStep(g, sampler, model, 64)
loader = BatchLoader(ids, 64, seed=5).forever()
step.calibrate(loader, steps=3)
step.capture(loader, warmup=1)
print(variant, "captured OK")
step(loader)
print(variant, "replayed OK", step.losses)
