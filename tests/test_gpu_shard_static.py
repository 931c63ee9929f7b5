"""GPU suite: the STATIC-SHAPE sharded sampler / train step (bliss_gnn_amd/shard_static.py: dense exchange, sizes on the device,
no host sync inside a step) against the routed, eager one of bliss_gnn_amd/shard.py -- which tests/test_gpu_shard.py pins to
the oracle in keyed mode.  One and two ranks (two ranks share the box's one GPU, gloo carries the collectives).

1. sampler: kept lists, inclusion probabilities, block edges / Hajek weights / q_ij, candidate counts, Poisson scales and the
   evolving EXP3 rows, bit for bit, over three steps;
2. the sampler replayed from a HIP graph (world of one rank) gives the same blocks as its eager enqueue;
3. the static train step == the eager sharded train step (predictions within bf16 tolerances, same kept sets)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu

from test_gpu_shard import BATCH, CLASSES, ETA, F, FAN, SEED, STEPS, V, _problem, _spawn   # noqa: E402


def _bits(t):
    return (t.view(torch.int16).to(torch.int32) & 0xFFFF)


def _block_content(b, S=None, K=None, B=None):
    from test_shard_gloo import block_records
    nid = b.srcdata["_ID"].long().cpu()
    S = b.num_dst_nodes() if S is None else S
    K = b.num_src_nodes() if K is None else K
    B = b.num_edges() if B is None else B
    dst_nid = nid[b.dst_pos.long().cpu()[:S]]
    rec = block_records(nid[:K], dst_nid, b.src.cpu()[:B], b.dst.cpu()[:B], b.edata["_ID"].cpu()[:B], b.edata["edge_weights"].cpu()[:B],
                        b.edata["q_ij"].cpu()[:B])
    return dict(rec=rec, kept=nid[:K].tolist(), prob=_bits(b.srcdata["node_prob"].cpu()[:K]).tolist(), dst=dst_nid.tolist())


def _sampler_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from bliss_gnn_amd import shard as sh
    from bliss_gnn_amd import shard_static as ss
    ip, ix, ei, batches, embed, _, _ = _problem()
    bounds = sh.partition_by_in_edges(ip, world)
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, rank, device=dev)
    eager = sh.ShardedPoissonBanditSampler(g, FAN, eta=ETA, seed=SEED)
    static = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
    embed = embed.to(dev)
    L = len(FAN)
    problems = []
    for step, seeds in enumerate(batches):
        _, _, eb = eager.sample_blocks(seeds.to(dev), step=step)
        _, _, sb = static.sample_blocks(seeds.to(dev), step=step)
        for l in range(L):
            n = L - 1 - l
            sz = static.sizes[n]
            want, got = _block_content(eb[l]), _block_content(sb[l], sz["S"], sz["K"], sz["B"])
            for k in ("kept", "prob", "dst", "rec"):
                if want[k] != got[k]:
                    problems.append(f"step {step} block {l}: {k} differs")
            if (eager.trace[n]["C"], eager.trace[n]["scale"]) != (static.trace[n]["C"], static.trace[n]["scale"]) and not eager.trace[n]["scale"][1]:
                problems.append(f"step {step} layer {n}: C / scale {eager.trace[n]['C'], eager.trace[n]['scale']} vs {static.trace[n]['C'], static.trace[n]['scale']}")
            eb[l].srcdata["embed_norm"] = embed[l][eb[l].srcdata["_ID"].long()]
            sb[l].srcdata["embed_norm"] = embed[l][sb[l].srcdata["_ID"].long().clamp(0, V - 1)]
        eager.exp3(eb)
        static.exp3(sb)
        eager.check_errors(); static.check_errors()
        if not torch.equal(eager.ops.w_pos, static.ops.w_pos):
            problems.append(f"step {step}: EXP3 rows differ in {int((eager.ops.w_pos != static.ops.w_pos).sum())} entries")
    torch.save(dict(rank=rank, problems=problems), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_static_sharded_sampler_equals_the_routed_one(cuda, world):
    with tempfile.TemporaryDirectory() as outdir:
        res = _spawn(_sampler_worker, world, outdir)
    for r in res:
        assert r["problems"] == [], r["problems"][:6]
