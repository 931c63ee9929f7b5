"""Step time over a long run, in windows (does the renormalisation pass start to run as the EXP3 weights evolve?)."""
import sys, time, torch
sys.path.insert(0, '.')
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader, PipelinedTrainStep
from bliss_gnn_amd import roofline
dev = torch.device('cuda:0')
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'reddit']
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
feats, labels, train_nid = node_data(cfg["num_nodes"], cfg["feat"], cfg["classes"], cfg["n_train"], seed=1, device=dev)
g = bg.Graph(ip, ix, ei, ndata={"features": feats, "labels": labels}); g.edata["w"] = bg.normalized_edata(g)
sampler = bg.PoissonBanditLadiesSampler(cfg["fanouts"], eta=0.1)
torch.manual_seed(1234)
model = SAGE(cfg["feat"], 256, cfg["classes"], 3, torch.relu, 0.1).to(dev).bfloat16()
loader = BatchLoader(train_nid, cfg["batch"], seed=2).forever()
step = PipelinedTrainStep(g, sampler, model, cfg["batch"])
step.calibrate(loader, steps=8); step.capture(loader, warmup=2, tune_gemm=True)
step.run(loader, 10)
for w in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sizes = step.run(loader, 200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    skipped = [int(x) for x in ((sampler._scratch[:, 0] >> 16) & 1).tolist()]
    print("steps %5d-%5d: %.3f ms/step   last norms %s  last pass skipped per layer %s  maxB %d" % (
        w * 400, w * 400 + 399, 1e3 * dt / 400, [float(x) for x in sampler._norms.float().tolist()], skipped, max(s[0]["B"] for s in sizes)), flush=True)
