"""debug: static sharded step, eager vs eager vs graphed (world 1, RCCL)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from test_gpu_shard import BATCH, CLASSES, ETA, F, FAN, SEED, V, _problem
cuda = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = "29745"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
from bliss_gnn_amd import shard as sh, shard_static as ss
from bliss_gnn_amd.model import SAGE
ip, ix, ei, _, _, feats, labels = _problem()
bounds = sh.partition_by_in_edges(ip, 1)
gen = torch.Generator().manual_seed(11)
batches = [torch.randperm(V, generator=gen)[:BATCH].to(torch.int32).to(cuda) for _ in range(7)]
for mode in ("eager", "graph"):
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, 0, device=cuda, ndata={"features": feats, "labels": labels})
    sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
    torch.manual_seed(0)
    model = SAGE(F, 32, CLASSES, 3, torch.relu, 0.0).to(cuda).bfloat16()
    step = ss.StaticShardedTrainStep(g, sampler, model, BATCH, lr=0.002)
    it = iter(batches)
    out = []
    def snap(tag):
        loss, sizes = step.finish()
        out.append((tag, round(loss, 5), [s["K"] for s in sizes], [s["B"] for s in sizes], float(sum(p.float().abs().sum() for p in model.parameters())),
                    int(sampler.ops.w_pos.view(torch.int16).long().sum()), [(t["C"], t["scale"]) for t in sampler.trace], [s["S"] for s in sizes],
                    int(sampler.ops.eng._bin_buffers()["cursor"][sampler.ops.eng.n_bins]), int(sampler._bufs["step"])))
    if mode == "graph":
        step.capture(it, warmup=2); snap("cap")
    else:
        for i in range(3):
            step(next(it)); snap("e%d" % i)
    for b in it:
        step(b); snap("s")
    print(mode)
    for o in out: print("   ", o)
    step.close()
dist.destroy_process_group()
