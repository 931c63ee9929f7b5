"""Multi-GPU replicas: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference is single-device (no collectives anywhere).  Round-1 scaling
design, chosen because every BASELINE graph fits one 288 GB part many times over: every rank holds
the whole graph and bandit state, samples its OWN batch (weak scaling), and two small exchanges per
step keep the replicas in lock-step:

  * gradients: ONE flat bf16 bucket (~0.45 M parameters, < 1 MB) all-reduced and averaged.  The
    message is far below the xGMI bandwidth knee, so a single latency-bound collective beats
    per-tensor or ring-pipelined ones;
  * EXP3: each rank computes the multiplicative factors of its own blocks without applying them
    (bliss_exp3_update(apply=0)); (position, factor) lists are all-gathered (padded to the longest,
    ~6 B per sampled edge) and EVERY rank applies ALL lists in rank order, then renormalises -- so
    the bf16 weight rows stay bit-identical on every GPU.  In the static-shape step the lists of all
    ranks and blocks are applied by one launch (bliss_exp3_apply_ranks: grid barrier between ranks).

The collective logic below is backend-agnostic and covered by world_size-2 gloo tests on CPU.
"""
import torch
import torch.distributed as dist


def want_hw_queues(n=8):
    """Call before the process's first GPU call.  The pipelined train loop keeps four streams busy at once (critical
    chain, backward pass, early-layer blocks, random-number generator); RCCL adds its own, and HIP multiplexes all streams
    onto 4 hardware queues by default.  Two busy streams on one queue serialise -- measured on a world of one rank: the
    generator landed on the backward pass's queue and the step took 1.40 ms instead of 0.85.  GPU_MAX_HW_QUEUES is read when
    the HIP runtime initialises.  (Not the default for single-GPU runs: with 8 queues a step recorded as ONE graph --
    GraphedTrainStep -- replays at 2.2 ms instead of 1.2.)"""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(n)))


def broadcast_parameters(model, src=0):
    for p in model.parameters():
        dist.broadcast(p.data, src)
    for b in model.buffers():
        dist.broadcast(b.data, src)


def allreduce_gradients(model):
    """Average all gradients with one flat-bucket all-reduce."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])                  # one kernel
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    views, off = [], 0
    for g in grads:
        n = g.numel()
        views.append(flat[off:off + n].view_as(g))
        off += n
    torch._foreach_copy_(grads, views)                                 # one multi-tensor launch instead of one copy per parameter


def gather_updates(pos, factor):
    """All-gather ragged (pos int32 [n], factor bf16 [n]) lists.  Returns, identically on every rank,
    ``[(pos_r, factor_r) for r in range(world)]``."""
    world = dist.get_world_size()
    n = torch.tensor([pos.numel()], dtype=torch.int64, device=pos.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    # one int32 payload: position in the low word, factor bits in a second int32 (bf16 is not a gloo dtype)
    buf = torch.zeros(2, m, dtype=torch.int32, device=pos.device)
    buf[0, : pos.numel()] = pos
    buf[1, : pos.numel()] = factor.view(torch.int16).to(torch.int32)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [(o[0, :s].contiguous(), o[1, :s].to(torch.int16).view(torch.bfloat16).contiguous()) for o, s in zip(out, sizes)]


def exp3_all_ranks(sampler, mfgs, g):
    """sampler.exp3 for replicas: compute locally, exchange, apply every rank's updates in rank order."""
    factors = [torch.empty(m.num_edges(), dtype=torch.bfloat16, device=g.device) for m in mfgs]
    sampler.exp3(mfgs, g, apply=False, factors=factors)
    for idx, mfg in enumerate(mfgs):
        for pos_r, fac_r in gather_updates(mfg.pos, factors[idx]):
            sampler.apply_updates(idx, pos_r, fac_r, g)
        sampler.normalize(idx, g)


def exp3_all_ranks_static(sampler, mfgs, g):
    """The same exchange for capacity-padded blocks with the true edge counts on the device: fixed-size messages, no
    host round trip -- the whole step, collectives included, can be recorded into one HIP graph.

    ONE all-gather per step: every rank packs the positions, factors and true counts of all its blocks into one int32
    buffer (the update kernel writes the factors straight into it); the exchange is latency-bound (a few MB over xGMI), so
    one collective instead of two per layer is what counts."""
    world = dist.get_world_size()
    L = len(mfgs)
    # per block: how many edges are exchanged (the engine's X capacity: tighter than the block's own capacity, these bytes
    # cross xGMI every step; a block with more edges raises error bit 8 on every rank)
    caps = [min(int(m.src.numel()), int(getattr(m, "_xcap", m.src.numel()))) for m in mfgs]
    offs = [sum(caps[:i]) for i in range(L)]
    tot = sum(caps)
    n_fac = (tot + 1) // 2                                   # bf16 factors, two per int32 word
    n = tot + n_fac + L
    n_pad = (n + 3) // 4 * 4
    buf = torch.empty(n_pad, dtype=torch.int32, device=g.device)
    fac_all = buf[tot:tot + n_fac].view(torch.bfloat16)      # [2 * n_fac] bf16 view of the same storage
    factors = [fac_all[offs[i]:offs[i] + caps[i]] for i in range(L)]
    sampler.exp3(mfgs, g, apply=False, factors=factors, bounds=caps)      # rewards + factors, nothing applied yet
    # positions and true counts (LayerCounts::B) of all blocks: one launch, only the entries that exist
    import ctypes as C
    from . import _lib
    pk = _lib.PackLists()
    for i, m in enumerate(mfgs):
        pk.pos[i], pk.n_dev[i] = m.pos.data_ptr(), m._counts_dev.data_ptr() + 16
        pk.pos_off_words[i], pk.count_off_words[i], pk.bound[i] = offs[i], tot + n_fac + i, caps[i]
    pk.n_blocks = L
    _lib.check(_lib.lib.bliss_pack_lists(C.byref(pk), buf.data_ptr(), torch.cuda.current_stream().cuda_stream), "bliss_pack_lists")
    gath = torch.empty(world * n_pad, dtype=torch.int32, device=g.device)
    dist.all_gather_into_tensor(gath, buf)
    # every rank's lists, rank after rank (rank order on every rank: the rows stay bit-identical), all blocks: ONE launch
    sampler.apply_updates_ranks(list(range(L)), gath, n_pad, world, offs, [2 * tot + o for o in offs],
                                [tot + n_fac + i for i in range(L)], caps)
    for i in range(L):
        sampler.normalize(i, g)
