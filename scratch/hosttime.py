"""Host-side enqueue time of the pipelined loop vs its GPU time: is the host keeping ahead of the device?"""
import sys, time, torch
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
step.calibrate(loader, steps=8); step.capture(loader, warmup=2, tune_gemm=True)
step.run(loader, 50)
torch.cuda.synchronize()
import cProfile, pstats
for rep in range(3):
    t0 = time.perf_counter()
    step.run(loader, 100)
    t1 = time.perf_counter()
    print("run(100 pairs): %.3f ms/step wall" % (1e3 * (t1 - t0) / 200))
# host-only cost of one pair's enqueue (device idle at the start, nothing waits)
torch.cuda.synchronize()
eng = sampler._engine
t0 = time.perf_counter()
for k in range(4):
    step._replay(first_chain=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue of 4 pairs: %.3f ms host per step; drained after %.3f ms per step" % (1e3 * (t1 - t0) / 8, 1e3 * (t2 - t0) / 8))
pr = cProfile.Profile(); pr.enable()
step.run(loader, 50)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
