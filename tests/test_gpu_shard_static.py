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


def _train_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from bliss_gnn_amd import shard as sh
    from bliss_gnn_amd import shard_static as ss
    from bliss_gnn_amd.model import SAGE
    ip, ix, ei, batches, _, feats, labels = _problem()
    bounds = sh.partition_by_in_edges(ip, world)
    g = sh.GraphShard.from_global(ip, ix, ei, bounds, rank, device=dev, ndata={"features": feats, "labels": labels})
    per_rank = BATCH // world
    out = {}
    def my_batch(si):
        # every rank contributes the same number of seeds it owns (static shapes)
        gen = torch.Generator().manual_seed(100 + 7 * si + rank)
        return (torch.randperm(g.hi - g.lo, generator=gen)[:per_rank] + g.lo).to(torch.int32).to(dev)

    for kind in ("eager", "static", "pipelined"):
        torch.manual_seed(0)
        model = SAGE(F, 32, CLASSES, 3, torch.relu, 0.0).to(dev).bfloat16()
        if kind == "eager":
            sampler = sh.ShardedPoissonBanditSampler(g, FAN, eta=ETA, seed=SEED)
            step = sh.ShardedTrainStep(g, sampler, model, lr=0.002)
        elif kind == "static":
            sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
            step = ss.StaticShardedTrainStep(g, sampler, model, per_rank, lr=0.002)
        else:                       # the two-stream loop, launched kernel by kernel (gloo): its backward stream has its own group
            sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
            step = ss.PipelinedShardedTrainStep(g, sampler, model, per_rank, lr=0.002)
            step.prime(my_batch(0))
            losses = []
            for si in range(len(batches)):
                step(my_batch(si + 1))
                losses.append(step.finish()[0])
            sampler.check_errors()
            out[kind] = dict(losses=losses, params=[p.detach().float().cpu() for p in model.parameters()],
                             w=sampler.ops.w_pos.cpu().view(torch.int16))
            continue
        losses, preds, kept = [], [], []
        for si in range(len(batches)):
            mine = my_batch(si)
            if kind == "eager":
                losses.append(float(step(mine)))
                b = step.last["mfgs"][-1]
                preds.append((b.dstdata["_ID"].cpu(), step.last["pred"].detach().float().cpu()))
                kept.append([m.srcdata["_ID"].cpu().tolist() for m in step.last["mfgs"]])
            else:
                step(mine)
                loss, sizes = step.finish()
                losses.append(loss)
                blocks = step.last["mfgs"]
                n0 = sizes[0]["S"]
                preds.append((blocks[-1].dstdata["_ID"].cpu()[:n0], step.last["pred"].float().cpu()[:n0]))
                kept.append([m.srcdata["_ID"].cpu()[:sizes[len(FAN) - 1 - l]["K"]].tolist() for l, m in enumerate(blocks)])
            sampler.check_errors()
        out[kind] = dict(losses=losses, preds=preds, kept=kept, params=[p.detach().float().cpu() for p in model.parameters()],
                         w=sampler.ops.w_pos.cpu().view(torch.int16))
    torch.save(dict(rank=rank, **out), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_static_sharded_train_step_follows_the_eager_one(cuda, world):
    """Step 0 starts from identical parameters and EXP3 rows: same kept lists, predictions and loss within bf16 tolerances (the
    capacity-padded GEMMs may tile differently); the input-most EXP3 row (a function of the FEATURES' norms) identical.  Later
    steps: both paths keep training (finite, close losses) -- their optimisers differ (one-launch bf16 Adam vs torch's)."""
    with tempfile.TemporaryDirectory() as outdir:
        res = _spawn(_train_worker, world, outdir)
    for r in res:
        e, s = r["eager"], r["static"]
        assert e["kept"][0] == s["kept"][0]
        ref = dict(zip(e["preds"][0][0].tolist(), e["preds"][0][1]))
        got = dict(zip(s["preds"][0][0].tolist(), s["preds"][0][1]))
        assert set(ref) == set(got)
        for nid, row in got.items():
            tol = 4 * 2.0 ** -8 * max(1.0, float(ref[nid].abs().max()))
            assert torch.allclose(row, ref[nid], rtol=3e-2, atol=tol)
        assert abs(e["losses"][0] - s["losses"][0]) <= 2e-2 * max(1.0, abs(e["losses"][0]))
        for a, b in zip(e["losses"], s["losses"]):
            assert a == a and b == b and abs(a - b) <= 0.15 * max(1.0, abs(a))
        assert all(torch.isfinite(p).all() for p in s["params"])
        pl = r["pipelined"]                                     # the two-stream loop == the one-stream static step, bit for bit
        assert pl["losses"] == s["losses"] and torch.equal(pl["w"], s["w"])
        assert all(torch.equal(a, b) for a, b in zip(pl["params"], s["params"]))
    if world == 2:
        assert all(torch.equal(a, b) for a, b in zip(res[0]["static"]["params"], res[1]["static"]["params"]))   # replicas of the parameters stay in step
        assert res[0]["static"]["losses"] == res[1]["static"]["losses"]


def test_static_sharded_step_replays_from_one_graph(cuda):
    """World of one rank over RCCL: the whole step (sampler with its dense all-reduces, halo all-reduces, model, Adam, EXP3) is
    recorded into ONE HIP graph; replaying it trains exactly like launching it kernel by kernel."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29743"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        from bliss_gnn_amd import shard as sh
        from bliss_gnn_amd import shard_static as ss
        from bliss_gnn_amd.model import SAGE
        ip, ix, ei, _, _, feats, labels = _problem()
        bounds = sh.partition_by_in_edges(ip, 1)
        gen = torch.Generator().manual_seed(11)
        batches = [torch.randperm(V, generator=gen)[:BATCH].to(torch.int32).to(cuda) for _ in range(9)]
        outs = []
        for graphed in (False, True):
            g = sh.GraphShard.from_global(ip, ix, ei, bounds, 0, device=cuda, ndata={"features": feats, "labels": labels})
            sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
            torch.manual_seed(0)
            model = SAGE(F, 32, CLASSES, 3, torch.relu, 0.0).to(cuda).bfloat16()
            step = ss.StaticShardedTrainStep(g, sampler, model, BATCH, lr=0.002)
            it = iter(batches)
            losses = []
            step.calibrate(it, steps=2)                     # capacities from observed sizes (both runs: the same two batches)
            assert sampler.ops.fixed_caps is not None and sampler.ops.eng.caps[0]["K"] <= 2 * (FAN[-1] + BATCH) + 64
            if graphed:
                step.capture(it, warmup=2)                  # batches 0, 1 eagerly, batch 2 by the first replay
                assert step.graph is not None
            else:
                for _ in range(3):
                    step(next(it))
            for b in it:
                step(b)
                losses.append(step.finish()[0])
            sampler.check_errors()
            outs.append((losses, [p.detach().float().cpu() for p in model.parameters()], sampler.ops.w_pos.cpu().view(torch.int16).clone()))
            step.close()
        assert outs[0][0] == outs[1][0]
        assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
        assert torch.equal(outs[0][2], outs[1][2])
    finally:
        dist.destroy_process_group()


def test_pipelined_sharded_loop_trains_like_the_one_stream_step(cuda):
    """World of one rank over RCCL.  PipelinedShardedTrainStep (F(t) -> X(t) -> S(t+1) on one stream, loss / backward / Adam of batch t
    on a second) leaves the losses, parameters and EXP3 rows of StaticShardedTrainStep bit for bit -- launched kernel by kernel, and
    replayed from its six graphs."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29751"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        from bliss_gnn_amd import shard as sh
        from bliss_gnn_amd import shard_static as ss
        from bliss_gnn_amd.model import SAGE
        ip, ix, ei, _, _, feats, labels = _problem()
        bounds = sh.partition_by_in_edges(ip, 1)
        gen = torch.Generator().manual_seed(13)
        batches = [torch.randperm(V, generator=gen)[:BATCH].to(torch.int32).to(cuda) for _ in range(8)]
        outs = {}
        for kind in ("one-stream", "pipelined", "pipelined-graphs"):
            g = sh.GraphShard.from_global(ip, ix, ei, bounds, 0, device=cuda, ndata={"features": feats, "labels": labels})
            sampler = ss.DenseShardedSampler(g, FAN, eta=ETA, seed=SEED)
            torch.manual_seed(0)
            model = SAGE(F, 32, CLASSES, 3, torch.relu, 0.0).to(cuda).bfloat16()
            losses = []
            if kind == "one-stream":
                step = ss.StaticShardedTrainStep(g, sampler, model, BATCH, lr=0.002)
                for b in batches[:7]:                       # trains batches 0 .. 6
                    step(b)
                    losses.append(step.finish()[0])
            else:
                step = ss.PipelinedShardedTrainStep(g, sampler, model, BATCH, lr=0.002)
                it = iter(batches)
                if kind == "pipelined-graphs":
                    step.capture(it, warmup=2)              # prime(0), call(1), call(2): batches 0, 1 trained kernel by kernel
                    assert step.graph
                else:
                    step.prime(next(it))
                for b in it:                                # every call trains the batch sampled one call earlier: 0 .. 6 in the end
                    step(b)
                    losses.append(step.finish()[0])
            sampler.check_errors()
            outs[kind] = (losses, [p.detach().float().cpu() for p in model.parameters()], sampler.ops.w_pos.cpu().view(torch.int16).clone())
            step.close()
        assert outs["pipelined"][0] == outs["one-stream"][0]
        assert outs["pipelined-graphs"][0] == outs["one-stream"][0][2:]
        for kind in ("pipelined", "pipelined-graphs"):
            assert all(torch.equal(a, b) for a, b in zip(outs[kind][1], outs["one-stream"][1])), kind
            assert torch.equal(outs[kind][2], outs["one-stream"][2]), kind
    finally:
        dist.destroy_process_group()


def test_row_kernels_place_and_take(cuda):
    """bliss_shard_place_rows / bliss_shard_take_rows against the torch form they replace (zeros + index_copy_; index_select +
    .to(bfloat16)): same bits, including the +0 padding rows and the round-to-nearest-even of the fp32 gradient buffer."""
    from bliss_gnn_amd import shard_static as ss
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    for cap_s, n, n_rows, D in ((64, 37, 500, 256), (300, 300, 301, 602), (128, 0, 77, 2), (50, 1, 9, 130)):
        pos_true = torch.sort(torch.randperm(n_rows, generator=gen)[:n]).values.to(torch.int32)
        pos = torch.zeros(cap_s, dtype=torch.int32)
        pos[:n] = pos_true                                        # (the padding entries point at row 0, as the sampler leaves them)
        pos, n_dev = pos.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
        h = torch.randn(cap_s, D, generator=gen).to(dev).bfloat16()
        out = ss._place_rows(h, pos, n_dev, n_rows)
        want = torch.zeros(n_rows, D, dtype=torch.bfloat16, device=dev)
        want[pos_true.long().to(dev)] = h[:n]
        assert torch.equal(out.view(torch.int16), want.view(torch.int16))
        for src in (torch.randn(n_rows, D, generator=gen).to(dev).bfloat16(), (torch.randn(n_rows, D, generator=gen) * 3).to(dev)):
            got = ss._take_rows(src, pos, n_dev, cap_s)
            ref = torch.zeros(cap_s, D, dtype=torch.bfloat16, device=dev)
            ref[:n] = src[pos_true.long().to(dev)].to(torch.bfloat16)
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # through autograd: the same values and gradients as the index forms
    cap_s, n, n_rows, D = 96, 70, 400, 64
    pos_true = torch.sort(torch.randperm(n_rows, generator=gen)[:n]).values
    pos = torch.zeros(cap_s, dtype=torch.int32); pos[:n] = pos_true.to(torch.int32)
    pos, n_dev = pos.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
    x = torch.randn(n_rows, D, generator=gen).to(dev).bfloat16()
    w = torch.randn(cap_s, D, generator=gen).to(dev).bfloat16()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = ss._TakeRowsHip.apply(xa, pos, n_dev)
    yb = ss._TakeRows.apply(xb, pos.long())
    assert torch.equal(ya[:n], yb[:n]) and not ya[n:].any()
    mask = (torch.arange(cap_s, device=dev) < n)[:, None]
    (ya * w).sum().backward(); (torch.where(mask, yb, 0.0) * w).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29747"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        ha, hb = w.clone().requires_grad_(True), w.clone().requires_grad_(True)
        za = ss._PlaceAndReduceHip.apply(ha, pos, n_dev, n_rows, None, None)
        idx = torch.where(torch.arange(cap_s, device=dev) < n, pos.long(), n_rows)
        zb = ss._PlaceAndReduce.apply(hb, idx, n_rows, None, None)
        assert torch.equal(za.view(torch.int16), zb.view(torch.int16))
        (za.float() * x.float()).sum().backward(); (zb.float() * x.float()).sum().backward()
        assert torch.equal(ha.grad, hb.grad)
    finally:
        dist.destroy_process_group()


def test_masked_cross_entropy_kernel(cuda):
    """bliss_cross_entropy_masked against a plain fp32 torch reference: sum over the first *n rows of CE(bf16(a + b), label) / denom,
    gradient (softmax - onehot) / denom on those rows (within one bf16 rounding), exactly +0 on the padding rows -- which may
    carry destination ids outside this rank's range."""
    from bliss_gnn_amd import _lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    cap, n, C, lo, n_table, denom = 96, 61, 41, 1000, 500, 512.0
    a = torch.randn(cap, C, generator=gen).to(dev).bfloat16()
    b = torch.randn(cap, C, generator=gen).to(dev).bfloat16()
    table = torch.randint(0, C, (n_table,), generator=gen).to(dev)
    ids = torch.randint(lo, lo + n_table, (cap,), generator=gen).to(torch.int32)
    ids[n:] = 7                                                   # (padding: another rank's node)
    ids = ids.to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    state = torch.zeros(2, dtype=torch.int32, device=dev)
    for second in (b, None):
        dx = torch.full((cap, C), 9.0, dtype=torch.bfloat16, device=dev)
        rows = torch.empty(cap, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.bliss_cross_entropy_masked(a.data_ptr(), a.stride(0), 0 if second is None else second.data_ptr(),
                                                       0 if second is None else second.stride(0), table.data_ptr(), n_table, ids.data_ptr(), lo,
                                                       cap, n_dev.data_ptr(), denom, C, rows.data_ptr(), dx.data_ptr(), dx.stride(0),
                                                       loss.data_ptr(), state.data_ptr(), state.data_ptr() + 4,
                                                       torch.cuda.current_stream().cuda_stream), "bliss_cross_entropy_masked")
        torch.cuda.synchronize()
        x = (a if second is None else (a.float() + second.float()).bfloat16()).float()[:n].requires_grad_(True)
        y = table[(ids[:n].long() - lo)]
        ref = torch.nn.functional.cross_entropy(x, y, reduction="sum") / denom
        ref.backward()
        assert int(state[1]) == 0 and int(state[0]) == 0
        assert abs(float(loss) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
        assert not dx[n:].view(torch.int16).any()
        g, gr = dx[:n].float(), x.grad
        assert torch.allclose(g, gr, rtol=2.0 ** -7, atol=2.0 ** -7 / denom)
