"""debug: world-2 static vs eager sharded forward, per-layer embed norms (run: python scratch/ss_debug.py)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from test_gpu_shard import BATCH, CLASSES, ETA, F, FAN, SEED, V, _problem, _spawn

def worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    from bliss_gnn_amd import shard as sh, shard_static as ss
    from bliss_gnn_amd.model import SAGE
    ip, ix, ei, batches, _, feats, labels = _problem()
    bounds = sh.partition_by_in_edges(ip, world)
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, rank, device=dev, ndata={"features": feats, "labels": labels})
    per_rank = BATCH // world
    gen = torch.Generator().manual_seed(100 + rank)
    mine = (torch.randperm(g.hi - g.lo, generator=gen)[:per_rank] + g.lo).to(torch.int32).to(dev)
    res = {}
    for kind in ("eager", "static"):
        torch.manual_seed(0)
        model = SAGE(F, 32, CLASSES, 3, torch.relu, 0.0).to(dev).bfloat16()
        if kind == "eager":
            sampler = sh.ShardedPoissonBanditSampler(g, FAN, eta=ETA, seed=SEED)
            step = sh.ShardedTrainStep(g, sampler, model, lr=0.002)
            step(mine)
            res[kind] = [(b.srcdata["_ID"].cpu(), b.srcdata["embed_norm"].float().cpu(), b.dst_pos.cpu(), b.num_edges()) for b in step.last["mfgs"]] + [step.last["pred"].float().cpu()]
        else:
            sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
            step = ss.StaticShardedTrainStep(g, sampler, model, per_rank, lr=0.002)
            step(mine); loss, sizes = step.finish()
            out = []
            for l, b in enumerate(step.last["mfgs"]):
                sz = sizes[len(FAN) - 1 - l]
                out.append((b.srcdata["_ID"].cpu()[:sz["K"]], b.srcdata["embed_norm"].float().cpu()[:sz["K"]], b.dst_pos.cpu()[:sz["S"]], sz["B"]))
            res[kind] = out + [step.last["pred"].float().cpu()[:sizes[0]["S"]]]
    for l in range(3):
        e, s = res["eager"][l], res["static"][l]
        print(rank, "block", l, "K", e[0].numel(), s[0].numel(), "ids equal", torch.equal(e[0], s[0].to(e[0].dtype)), "norm maxdiff", float((e[1] - s[1]).abs().max()),
              "n diff", int((e[1] != s[1]).sum()), "dst_pos equal", torch.equal(e[2].long(), s[2].long()), "B", e[3], s[3], flush=True)
        if (e[1] != s[1]).any():
            bad = torch.nonzero(e[1] != s[1]).flatten()[:8]
            print(rank, "   first diffs at", bad.tolist(), "ids", e[0][bad].tolist(), "owned", [(int(i) >= g.lo and int(i) < g.hi) for i in e[0][bad]], e[1][bad].tolist(), s[1][bad].tolist(), flush=True)
    print(rank, "pred maxdiff", float((res["eager"][3] - res["static"][3]).abs().max()), flush=True)
    torch.save(dict(rank=rank), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        _spawn(worker, 2, d)
