"""GPU suite: destination-range shards with the HIP kernels (bliss_gnn_amd/shard.py:_HipShardOps), two ranks sharing
the box's one GPU, gloo carrying the collectives through the host.

1. the sharded sampler + EXP3 update == the oracle in keyed mode, bit for bit (the CPU twin of this test,
   tests/test_shard_gloo.py, checks the same exchange logic with the oracle's arithmetic in place of the kernels);
2. the sharded SAGE step (halo gathers forward, reduce-scatter backward, gradient all-reduce) on 2 shards == on 1 shard:
   the blocks' kept lists come out in ascending node order for any number of shards, so predictions agree row by row."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu

V, E, FAN, BATCH, ETA, SEED, STEPS, F, CLASSES = 20000, 400000, [512, 256, 128], 64, 0.1, 7, 3, 48, 5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(V, E, seed=8)
    gen = torch.Generator().manual_seed(4)
    batches = [torch.randperm(V, generator=gen)[:BATCH].to(torch.int32) for _ in range(STEPS)]
    embed = (torch.rand(len(FAN), V, generator=gen) * 30).bfloat16()
    feats = torch.randn(V, F, generator=gen).bfloat16()
    labels = torch.randint(0, CLASSES, (V,), generator=gen)
    return ip, ix, ei, batches, embed, feats, labels


def _sampler_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from bliss_gnn_amd import shard as sh
    from test_shard_gloo import block_records
    ip, ix, ei, batches, embed, _, _ = _problem()
    bounds = sh.partition_by_in_edges(ip, world)
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, rank, device=dev)
    sampler = sh.ShardedPoissonBanditSampler(g, FAN, eta=ETA, seed=SEED)
    embed = embed.to(dev)
    out = []
    for step, seeds in enumerate(batches):
        inp, outp, blocks = sampler.sample_blocks(seeds.to(dev), step=step)
        recs = []
        for l, b in enumerate(blocks):
            nid = b.srcdata["_ID"]
            recs.append(dict(rec=block_records(nid.long().cpu(), nid.long()[b.dst_pos].cpu(), b.src.cpu(), b.dst.cpu(), b.edata["_ID"].cpu(),
                                               b.edata["edge_weights"].cpu(), b.edata["q_ij"].cpu()),
                             kept=nid.tolist(), prob=(b.srcdata["node_prob"].cpu().view(torch.int16).to(torch.int32) & 0xFFFF).tolist()))
            b.srcdata["embed_norm"] = embed[l][nid.long()]
        sampler.exp3(blocks)
        sampler.check_errors()
        out.append(dict(blocks=recs, trace=[(t["C"], t["scale"]) for t in sampler.trace],
                        w=(sampler.ops.w_pos.cpu().view(torch.int16).to(torch.int32) & 0xFFFF)))
    torch.save(dict(rank=rank, e0=int(ip[g.lo]), e1=int(ip[g.hi]), out=out), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(fn, world, outdir, *extra):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=fn, args=(r, world, port, outdir) + extra) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    return [torch.load(os.path.join(outdir, f"r{r}.pt"), weights_only=False) for r in range(world)]


def test_sharded_hip_sampler_matches_keyed_oracle(cuda):
    from oracle import bliss_oracle as bo
    from test_shard_gloo import block_records
    with tempfile.TemporaryDirectory() as outdir:
        res = _spawn(_sampler_worker, 2, outdir)
    ip, ix, ei, batches, embed, _, _ = _problem()
    og = bo.CSC(ip, ix, ei)
    edge_w = bo.normalized_edata(og)
    w = torch.ones(len(FAN), og.num_edges, dtype=torch.bfloat16)
    bits = lambda t: (t.view(torch.int16).to(torch.int32) & 0xFFFF)
    for step, seeds in enumerate(batches):
        fn = lambda n, nid: bo.keyed_uniform(SEED, step, n, nid)
        _, _, oblocks = bo.sample_blocks_bandit(og, seeds, FAN, w, ETA, uniform_fn=fn)
        for l, ob in enumerate(oblocks):
            want = block_records(ob.src_nid, ob.dst_nid, ob.src, ob.dst, ob.eid, ob.edge_weights, ob.q_ij)
            got = {}
            for r in res:
                blk = r["out"][step]["blocks"][l]
                assert not (set(blk["rec"]) & set(got))
                got.update(blk["rec"])
                assert dict(zip(blk["kept"], blk["prob"])) == dict(zip(ob.src_nid.tolist(), bits(ob.node_prob).tolist()))
            assert got == want
            n = len(FAN) - 1 - l
            for r in res:
                C_g, (c, all_one, iters) = r["out"][step]["trace"][n]
                assert C_g == ob.trace["cand_nid"].numel() and (all_one or (c == ob.trace["c"] and iters == ob.trace["iters"]))
        w, _ = bo.exp3(og, oblocks, w, edge_w, [embed[l][b.src_nid] for l, b in enumerate(oblocks)])
        w_pos = bits(w[:, og.eid.long()])
        for r in res:
            assert torch.equal(r["out"][step]["w"], w_pos[:, r["e0"]:r["e1"]])


def _train_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from bliss_gnn_amd import shard as sh
    from bliss_gnn_amd.model import SAGE
    ip, ix, ei, batches, _, feats, labels = _problem()
    bounds = sh.partition_by_in_edges(ip, world)
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, rank, device=dev, ndata={"features": feats, "labels": labels})
    sampler = sh.ShardedPoissonBanditSampler(g, FAN, eta=ETA, seed=SEED)
    torch.manual_seed(0)
    model = SAGE(F, 32, CLASSES, 3, torch.relu, 0.0).to(dev).bfloat16()
    step = sh.ShardedTrainStep(g, sampler, model, lr=0.002)
    losses, preds = [], []
    for seeds in batches:
        mine = seeds[(seeds >= g.lo) & (seeds < g.hi)].to(dev)            # the part of the batch this rank owns
        losses.append(step(mine))
        b = step.last["mfgs"][-1]
        preds.append((b.dstdata["_ID"].cpu(), step.last["pred"].detach().float().cpu()))
        sampler.check_errors()
    torch.save(dict(rank=rank, losses=losses, preds=preds, params=[p.detach().float().cpu() for p in model.parameters()],
                    w=sampler.ops.w_pos.cpu().view(torch.int16), e0=int(ip[g.lo]), e1=int(ip[g.hi])), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_train_step_two_shards_equal_one(cuda):
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        one = _spawn(_train_worker, 1, d1)[0]
        two = _spawn(_train_worker, 2, d2)
    assert all(torch.equal(a, b) for a, b in zip(two[0]["params"], two[1]["params"]))        # replicated parameters stay in step
    for s in range(STEPS):
        ref = dict(zip(one["preds"][s][0].tolist(), one["preds"][s][1]))
        n = 0
        for r in two:
            for nid, row in zip(r["preds"][s][0].tolist(), r["preds"][s][1]):
                # bf16 rows after three bf16 layers (and, from step 1 on, parameters that have drifted by bf16 roundings): a few
                # bf16 ulps of the ROW's scale -- the small logits of a row carry the absolute error of its large ones
                tol = 4 * 2.0 ** -8 * max(1.0, float(ref[nid].abs().max()))
                assert torch.allclose(row, ref[nid], rtol=3e-2, atol=tol)
                n += 1
        assert n == len(ref)
        assert abs(two[0]["losses"][s] - one["losses"][s]) <= 2e-2 * max(1.0, abs(one["losses"][s]))
        assert two[0]["losses"][s] == two[1]["losses"][s]
    # step 0 saw identical inputs: the EXP3 rows (a function of the blocks and the row norms of the inputs) agree wherever
    # the bf16 activations' norms agree -- for the input-most block (norms of the FEATURES) exactly
    w1 = one["w"]
    for r in two:
        assert torch.equal(r["w"][0], w1[0][r["e0"]:r["e1"]]) or (r["w"][0] != w1[0][r["e0"]:r["e1"]]).float().mean() < 0.02
