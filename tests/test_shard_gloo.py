"""CPU suite: the destination-range-sharded sampler (bliss_gnn_amd/shard.py) on world_size-2 (and 3) gloo.

The exchange logic under test is the product's (ShardedPoissonBanditSampler: partition, all-to-all of partial sums to the
source owners, histogram all-reduce, keyed draw at the owner, kept-list all-gather, block assembly, global renormalisation
of the EXP3 rows); the per-shard arithmetic is supplied by tests/shard_cpu_ops.py (the oracle's, on CPU tensors) because
the HIP kernels need a GPU -- tests/test_gpu_shard.py runs the same comparison with the HIP ops on the GPU box.

Checked against the single-process oracle in the same keyed mode (uniform of a candidate = f(seed, step, layer, node id)):
kept sets, P, block edges, Hajek weights, q_ij and the evolving EXP3 rows, bit for bit, over three steps."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

V, E, FAN, BATCH, ETA, SEED, STEPS = 3000, 40000, [96, 48, 24], 24, 0.1, 11, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(V, E, seed=5)
    gen = torch.Generator().manual_seed(3)
    batches = [torch.randperm(V, generator=gen)[:BATCH].to(torch.int32) for _ in range(STEPS)]
    embed = (torch.rand(len(FAN), V, generator=gen) * 30).bfloat16()          # stands in for ||h_j||, by (block, node id)
    return ip, ix, ei, batches, embed


def block_records(src_nid, dst_nid, src, dst, eid, w, q):
    """{(src node, dst node): (edge id, W~ bits, q bits)} -- numbering-independent content of a block."""
    bits = lambda t: (t.view(torch.int16).to(torch.int32) & 0xFFFF).tolist()
    s, d = src_nid[src.long()].tolist(), dst_nid[dst.long()].tolist()
    return {(a, b): (int(e), x, y) for a, b, e, x, y in zip(s, d, eid.tolist(), bits(w), bits(q))}


def _worker(rank, world, port, outdir, exchange="routed"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bliss_gnn_amd import shard as sh
    from shard_cpu_ops import OracleShardOps
    ip, ix, ei, batches, embed = _problem()
    bounds = sh.partition_by_in_edges(ip, world)
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, rank)
    ops = OracleShardOps(g, len(FAN), ETA)
    if exchange == "dense":         # shard_static.py: ONE dense int64 [2, |V|] all-reduce per layer instead of the three routed exchanges
        from bliss_gnn_amd import shard_static as ss
        sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED, ops=ops)
    else:
        sampler = sh.ShardedPoissonBanditSampler(g, FAN, eta=ETA, seed=SEED, ops=ops)
    out = []
    for step, seeds in enumerate(batches):
        inp, outp, blocks = sampler.sample_blocks(seeds, step=step)
        recs = []
        for l, b in enumerate(blocks):
            nid = b.srcdata["_ID"]
            eid_g = g.eid[b.pos.long()] if b.num_edges() else b.pos
            recs.append(dict(rec=block_records(nid.long(), nid.long()[b.dst_pos], b.src, b.dst, eid_g, b.edata["edge_weights"], b.edata["q_ij"]),
                             kept=nid.tolist(), prob=(b.srcdata["node_prob"].view(torch.int16).to(torch.int32) & 0xFFFF).tolist(),
                             dst=nid[b.dst_pos].tolist()))
            b.srcdata["embed_norm"] = embed[l][nid.long()]
        sampler.exp3(blocks)
        out.append(dict(blocks=recs, trace=[(t["C"], t["scale"]) for t in sampler.trace],
                        w=[(ops.w[l].view(torch.int16).to(torch.int32) & 0xFFFF) for l in range(len(FAN))]))
    torch.save(dict(rank=rank, lo=g.lo, hi=g.hi, e0=int(ip[g.lo]), e1=int(ip[g.hi]), out=out), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _oracle_run():
    from oracle import bliss_oracle as bo
    ip, ix, ei, batches, embed = _problem()
    og = bo.CSC(ip, ix, ei)
    edge_w = bo.normalized_edata(og)
    w = torch.ones(len(FAN), og.num_edges, dtype=torch.bfloat16)
    steps = []
    for step, seeds in enumerate(batches):
        fn = lambda n, nid: bo.keyed_uniform(SEED, step, n, nid)
        _, _, blocks = bo.sample_blocks_bandit(og, seeds, FAN, w, ETA, uniform_fn=fn)
        en = [embed[l][b.src_nid] for l, b in enumerate(blocks)]
        w, _ = bo.exp3(og, blocks, w, edge_w, en)
        steps.append((blocks, w.clone()))
    return og, steps


@pytest.mark.parametrize("world,exchange", [(2, "routed"), (3, "routed"), (2, "dense"), (3, "dense")])
def test_sharded_sampler_matches_keyed_oracle(world, exchange):
    port = _free_port()
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker, args=(r, world, port, outdir, exchange)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
        res = [torch.load(os.path.join(outdir, f"r{r}.pt"), weights_only=False) for r in range(world)]
    og, steps = _oracle_run()
    bits = lambda t: (t.view(torch.int16).to(torch.int32) & 0xFFFF)
    assert sum(r["hi"] - r["lo"] for r in res) == V and all(r["hi"] > r["lo"] for r in res)
    for step, (oblocks, ow) in enumerate(steps):
        for l, ob in enumerate(oblocks):
            want = block_records(ob.src_nid, ob.dst_nid, ob.src, ob.dst, ob.eid, ob.edge_weights, ob.q_ij)
            got, dsts = {}, []
            for r in res:
                blk = r["out"][step]["blocks"][l]
                assert not (set(blk["rec"]) & set(got))                                   # every edge lives on exactly one rank
                got.update(blk["rec"])
                dsts += blk["dst"]
                # the kept list is GLOBAL and identical on every rank: same set as the oracle's, same P per node
                assert blk["kept"] == res[0]["out"][step]["blocks"][l]["kept"]
                assert sorted(blk["kept"]) == sorted(ob.src_nid.tolist())
                assert dict(zip(blk["kept"], blk["prob"])) == dict(zip(ob.src_nid.tolist(), bits(ob.node_prob).tolist()))
                # seeds first (in the batch's order for the output-most block; a later layer's seed ORDER differs from the
                # oracle's first-appearance order by construction, its seed SET does not)
                assert sorted(blk["kept"][: ob.n_dst]) == sorted(ob.dst_nid.tolist())
                if l == len(FAN) - 1:
                    assert blk["kept"][: ob.n_dst] == ob.dst_nid.tolist()
            assert got == want                                                            # edges, edge ids, W~ and q_ij bits
            assert sorted(dsts) == sorted(ob.dst_nid.tolist())                            # destinations partitioned over the ranks
            # candidate count and Poisson scale: the oracle's, on every rank (sampling order n = L-1-l)
            n = len(FAN) - 1 - l
            for r in res:
                C_g, (c, all_one, iters) = r["out"][step]["trace"][n]
                assert C_g == ob.trace["cand_nid"].numel() and (all_one or (c == ob.trace["c"] and iters == ob.trace["iters"]))
        # the EXP3 rows after the step: every rank holds its columns' slice of the oracle's rows (by CSC position)
        ow_pos = bits(ow[:, torch.argsort(og.eid.long())] if False else ow[:, og.eid.long()])
        for r in res:
            for l in range(len(FAN)):
                assert torch.equal(r["out"][step]["w"][l], ow_pos[l, r["e0"]:r["e1"]])


def _gather_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bliss_gnn_amd import shard as sh
    n_total = 7
    pos = torch.tensor([0, 3, 4] if rank == 0 else [1, 2, 5, 6])
    rows = (torch.arange(pos.numel() * 2, dtype=torch.float32).reshape(-1, 2) + 10 * rank).requires_grad_()
    full = sh.gather_rows(rows, pos, n_total, world)
    # every rank weights the full matrix differently: a row's gradient is the SUM over the ranks that consumed it
    wgt = torch.arange(n_total * 2, dtype=torch.float32).reshape(n_total, 2) * (rank + 1)
    (full * wgt).sum().backward()
    torch.save(dict(full=full.detach(), grad=rows.grad, pos=pos), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_forward_and_reduce_scatter_backward():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_gather_worker, args=(r, world, port, outdir)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
        a, b = (torch.load(os.path.join(outdir, f"r{r}.pt"), weights_only=False) for r in range(world))
    assert torch.equal(a["full"], b["full"])
    assert torch.equal(a["full"][a["pos"]], torch.arange(6, dtype=torch.float32).reshape(-1, 2))
    assert torch.equal(a["full"][b["pos"]], torch.arange(8, dtype=torch.float32).reshape(-1, 2) + 10)
    wsum = torch.arange(14, dtype=torch.float32).reshape(7, 2) * 3                        # (1 + 2) x the base weights
    assert torch.equal(a["grad"], wsum[a["pos"]]) and torch.equal(b["grad"], wsum[b["pos"]])


def _halo_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bliss_gnn_amd import shard_static as ss
    n_total, F_ = 7, 4
    pos = torch.tensor([0, 3, 4] if rank == 0 else [1, 2, 5, 6])
    gen = torch.Generator().manual_seed(5 + rank)
    rows = (torch.randn(pos.numel(), F_, generator=gen) * 3).bfloat16().requires_grad_()      # (negative values too: -x, and the zeros stay +0)
    buf = ss._PlaceRows.apply(rows, pos, n_total)                                             # my rows at their positions, zeros elsewhere
    full = ss.halo_all_reduce(buf)
    wgt = (torch.arange(n_total * F_, dtype=torch.float32).reshape(n_total, F_) * (rank + 1)).bfloat16()
    (full.float() * wgt.float()).sum().backward()
    taken = ss._TakeRows.apply(full.detach().requires_grad_(), torch.tensor([2, 2, 0]))
    torch.save(dict(full=full.detach(), rows=rows.detach(), grad=rows.grad, pos=pos, taken=taken.detach()), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_all_reduce_is_exact_forward_and_sums_gradients_backward():
    """shard_static's halo exchange: the sum of the ranks' zero-padded buffers, sent as int32 words, is the full input bit for bit;
    the gradient of a row is the sum over the ranks that consumed it."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_halo_worker, args=(r, world, port, outdir)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
        a, b = (torch.load(os.path.join(outdir, f"r{r}.pt"), weights_only=False) for r in range(world))
    bits = lambda t: t.view(torch.int16)
    assert torch.equal(bits(a["full"]), bits(b["full"]))
    assert torch.equal(bits(a["full"][a["pos"]]), bits(a["rows"])) and torch.equal(bits(a["full"][b["pos"]]), bits(b["rows"]))
    wsum = (torch.arange(28, dtype=torch.float32).reshape(7, 4).bfloat16().float() + torch.arange(28, dtype=torch.float32).reshape(7, 4).mul(2).bfloat16().float())
    assert torch.allclose(a["grad"].float(), wsum[a["pos"]], rtol=2 ** -7) and torch.allclose(b["grad"].float(), wsum[b["pos"]], rtol=2 ** -7)
    assert torch.equal(bits(a["taken"]), bits(a["full"][torch.tensor([2, 2, 0])]))
