"""debug: static sharded step at Reddit scale, calibrate + graph: where does the first non-finite value appear?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = "29747"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from bliss_gnn_amd import shard as sh, shard_static as ss
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader
cfg = CONFIGS["reddit"]
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
feats, labels, train_nid = node_data(cfg["num_nodes"], cfg["feat"], cfg["classes"], cfg["n_train"], seed=1, device=dev, multilabel=False)
bounds = sh.partition_by_in_edges(ip, 1)
g = sh.GraphShard.from_global(ip, ix, ei, bounds, 0, device=dev, ndata={"features": feats, "labels": labels})
torch.manual_seed(1234)
p_drop = float(os.environ.get("PDROP", "0.1"))
model = SAGE(cfg["feat"], 256, cfg["classes"], 3, torch.relu, p_drop).to(dev).bfloat16()
sampler = ss.DenseShardedSampler(g, cfg["fanouts"], eta=0.1, seed=7)
step = ss.StaticShardedTrainStep(g, sampler, model, cfg["batch"], lr=0.002)
loader = BatchLoader(train_nid, cfg["batch"], shuffle=True, drop_last=True, seed=2).forever()
step.calibrate(loader, steps=4)
print("caps", [(c["S"], c["K"], c["B"]) for c in sampler.ops.eng.caps], flush=True)
step.capture(loader, warmup=2)
nosync = os.environ.get("NOSYNC", "0") == "1"
every = int(os.environ.get("EVERY", "1"))
for i in range(int(os.environ.get("NSTEPS", "150"))):
    step(next(loader))
    if nosync and (i + 1) % every:
        continue
    loss, sizes = float(step.loss_dev.item()), sampler.finish_nothrow()
    bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    bad_w = int((~torch.isfinite(sampler.ops.w_pos.float())).sum())
    pred = step.last["pred"]
    n0 = sizes[0]["S"]
    print(i, "loss", round(loss, 4), "sizes", [(z["S"], z["K"], z["B"]) for z in sizes], "bad params", bad_p, "bad w", bad_w,
          "pred finite (valid/all)", bool(torch.isfinite(pred[:n0]).all()), bool(torch.isfinite(pred).all()),
          "err", hex(int(sampler.ops.err.item())), hex(int(sampler._bufs["err"].item())), [hex(z["err"]) for z in sizes],
          "scratch", hex(int((sampler.ops.scratch[:, 0] >> 20).max().item())),
          "embed_norm finite", [bool(torch.isfinite(b.srcdata["embed_norm"].float()).all()) for b in step.last["mfgs"]], flush=True)
    if bad_p or bad_w or loss != loss:
        break
step.close(); dist.destroy_process_group()
