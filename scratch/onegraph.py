"""Experiment: the whole pipelined pair as ONE HIP graph (sampler and backward as parallel branches).  On ROCm 7.2 the
graph executor ran both branches on one hardware queue; DEBUG_HIP_FORCE_GRAPH_QUEUES may change that.  Prints ms/step."""
import os, sys, time, torch
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
step.calibrate(loader, steps=8)
step.capture(loader, warmup=2, tune_gemm=True)            # the normal six-graph capture (also warms everything up)
step.run(loader, 10)
torch.cuda.synchronize()
t0 = time.perf_counter(); step.run(loader, 50); torch.cuda.synchronize()
print("six graphs, two streams: %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / 100))
# one graph for the pair
step._load(loader)
gp = torch.cuda.CUDAGraph()
with torch.cuda.graph(gp):
    losses = step._pair()
def once():
    step.seeds2[1].copy_(next(loader)); step.seeds2[0].copy_(next(loader))
    step.sampler._engine.stage_rng_from_torch()
    gp.replay()
    step._finish_pair()
for _ in range(5): once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): once()
torch.cuda.synchronize()
print("one graph per pair (env DEBUG_HIP_FORCE_GRAPH_QUEUES=%s): %.3f ms/step (syncs per pair)" % (os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES"), 1e3 * (time.perf_counter() - t0) / 100))
