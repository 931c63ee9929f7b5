"""Which aten ops launch the small kernels of the backward stream: torch.profiler over one eager pair."""
import sys, torch
sys.path.insert(0, '.')
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader, PipelinedTrainStep
from torch.profiler import profile, ProfilerActivity
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
step.calibrate(loader, steps=4); step.sampler._engine.scratch_sets = 3; step.prime(next(loader))
step.eager_pair(loader); step.eager_pair(loader)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step.eager_pair(loader)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
seen = 0
for e in evs:
    ks = [k.name[:48] for k in e.kernels]
    if any(("elementwise" in k or "fill" in k.lower() or "copy" in k.lower() or "reduce_kernel" in k or "scatter_gather" in k) for k in ks):
        shapes = str(e.input_shapes)[:70]
        print("%-34s %-72s %s" % (e.name[:34], shapes, ks[0]))
        seen += 1
print("small-kernel ops:", seen)
