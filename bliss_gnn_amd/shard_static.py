"""Destination-range shards with STATIC SHAPES: the sampler and the train step of
``bliss_gnn_amd.shard`` with every exchange made dense, so that no size reaches the host inside a step and the step records
into a HIP graph, collectives included.

``shard.py`` routes what it exchanges: (source, partial sum) pairs to the sources' owners, a histogram all-reduce, an
all-gather of the kept lists -- three exchanges per sampling layer with data-dependent sizes, i.e. a host sync each.  Here:

  * one sampling layer = ONE all-reduce of a fixed shape.  Every rank expands the columns of the seeds it owns and reduces
    (q/sum q)^2 by source over its own edges (bandit_sampler.py:67-73: the same kernels, ``BLISS_MODE_PARTIALS``), then scatters
    these exact Q.44 partial sums into an ``int64 [|V|, 2]`` buffer (sum and touch mark side by side, + 2^32 for a seed) which is all-reduced
    (1.9 MB on the Reddit-like graph).  Integer sums: the result has the same bits for any number of shards and any reduction
    order.  Every rank then holds every source's sum and derives THE SAME candidate list (ascending node id), importances,
    histogram, Poisson scale (:391-401), keyed draw (:403-406, :422-424) and kept list -- replicated work instead of routed data;
  * a rank builds the block of ITS seeds over the global kept list (:269-339), as in ``shard.py``;
  * block inputs (halo): each rank writes the rows it owns at their positions of a zero ``[cap_K, F]`` buffer and the buffers
    are all-reduced as int32 words (x + 0 + .. + 0: exact, and an integer type both RCCL and gloo reduce); the backward of that
    gather is the all-reduce of the fp32 gradient buffer;
  * EXP3: owner-local updates, the L1 norm from the all-reduced exact row sums (``bliss_exp3_normalize_global``).

Parity: ``DenseShardedSampler`` produces the kept lists, probabilities, blocks and EXP3 rows of ``ShardedPoissonBanditSampler``
bit for bit (tests/test_gpu_shard_static.py; the CPU twin tests/test_shard_static_gloo.py runs the dense exchange with the
oracle's arithmetic under gloo on 2 and 3 ranks against the single-process keyed oracle).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .graph import NID
from .shard import HIST_BINS, ShardBlock, _HipShardOps, _all_reduce

SEED_MARK = 1 << 32


def _cap_kw():
    """Keyword arguments of every ``torch.cuda.graph`` capture here: with a process group alive, ProcessGroupNCCL's watchdog thread
    queries its events at any time, and a capture in the default "global" error mode turns such a query from ANOTHER thread into
    "operation not permitted when stream is capturing" (seen once captures became frequent: bench.py --dist shards aborted)."""
    try:
        import torch.distributed as _d
        return {"capture_error_mode": "thread_local"} if _d.is_available() and _d.is_initialized() else {}
    except Exception:                                            # noqa: BLE001
        return {}


# ------------------------------------------------------------------------------------------- the exchange, generic form
class DenseShardedSampler:
    """PoissonBanditLadiesSampler (bandit_sampler.py:369-425) over destination-range shards, dense exchange.

    ``ops`` given (tests: the oracle's arithmetic on CPU tensors; or ``shard._HipShardOps``): the layer is a sequence of tensor
    ops with true sizes (host syncs) -- the REFERENCE FORM of the exchange, used to test it.  ``ops=None``: the static-shape
    HIP path (``enqueue`` / ``sample_blocks``): capacity-sized buffers, sizes on the device, graph-capturable."""

    def __init__(self, shard, nodes_per_layer, eta=0.4, importance_sampling=True, seed=0, model="sage", group=None, ops=None,
                 fixed_caps=None):
        self.g, self.nodes_per_layer, self.eta = shard, list(nodes_per_layer), eta
        self.seed, self.group, self.step = int(seed), group, 0
        self.world, self.rank = shard.world, shard.rank
        self.delta = 0.01                                                   # bandit_sampler.py:233
        self._delta_f = float(torch.tensor(self.delta, dtype=torch.float32))
        self.static = ops is None
        self.ops = ops if ops is not None else _HipShardOps(shard, len(self.nodes_per_layer), eta, importance_sampling, model)
        if fixed_caps is not None:
            self.ops.fixed_caps = [dict(c) for c in fixed_caps]
        self.trace = []
        self._bufs = None
        self.bytes_per_step = 0

    # ---------------------------------------------------------------- reference form (true sizes, any device)
    def sample_blocks_generic(self, seeds_g, step=None):
        g, ops, grp = self.g, self.ops, self.group
        dev, V = seeds_g.device, int(g.indptr.numel() - 1)
        step = self.step if step is None else int(step)
        self.step = step + 1
        seeds_g = seeds_g.to(torch.int32).contiguous()
        L = len(self.nodes_per_layer)
        order = list(reversed(range(L)))
        fan = [self.nodes_per_layer[b] for b in order]
        if hasattr(ops, "set_caps"):
            ops.set_caps(int(seeds_g.numel()), fan)
        blocks, self.trace = [], []
        for n, layer in enumerate(order):
            S_g = int(seeds_g.numel())
            mine = (seeds_g >= g.lo) & (seeds_g < g.hi)
            seed_pos = torch.nonzero(mine).flatten()
            seeds_l = seeds_g[seed_pos].contiguous()
            ids, sums = ops.frontier_partials(n, layer, seeds_l)              # my seeds first, then the other touched sources
            dense = torch.zeros(2, V, dtype=torch.int64, device=dev)
            dense[0, ids.long()] = sums
            dense[1, ids.long()] = 1
            dense[1, seeds_l.long()] += SEED_MARK
            _all_reduce(dense, grp)                                           # THE exchange of this layer
            cand = torch.nonzero(dense[1]).flatten()
            is_seed = dense[1, cand] >= SEED_MARK
            p = ops.importance(dense[0, cand])
            hist = torch.bincount(p.view(torch.int16).long() & 0xFFFF, minlength=HIST_BINS)[:HIST_BINS]
            ops.scale(hist, int(cand.numel()), fan[n])
            cand = cand.to(torch.int32)
            P, keep = ops.keyed_select(cand, p, is_seed, self.seed, step, n)
            new = keep & ~is_seed
            kept_g = torch.cat([seeds_g, cand[new]]).contiguous()
            prob_g = torch.cat([torch.ones(S_g, dtype=torch.bfloat16, device=dev), P[new]])
            blk = ops.build_block(n, kept_g, prob_g, seed_pos)
            blk.edata["edge_weights"] = blk._edge_weights
            blk.edata["q_ij"] = blk._q
            blk.srcdata["node_prob"] = blk._node_prob
            self.trace.append(dict(cand=cand, p=p, P=P, C=int(cand.numel()), scale=ops.scale_result() if hasattr(ops, "scale_result") else None))
            blocks.insert(0, blk)
            seeds_g = kept_g
        return blocks[0].srcdata[NID], blocks[-1].dstdata[NID], blocks

    # ---------------------------------------------------------------- static form (HIP, no host sync)
    def _ensure_static(self, S0, fan):
        ops, eng, dev = self.ops, self.ops.eng, self.g.device
        ops.set_caps(S0, fan)
        L, V = len(fan), eng.V
        key = (S0, tuple(fan), tuple((c["S"], c["K"], c["B"]) for c in eng.caps))
        if self._bufs is not None and self._bufs["key"] == key:
            return self._bufs
        nb = V // 1024 + 4
        b = dict(key=key, counts=torch.zeros(L * 10, dtype=torch.int32, device=dev), rec=torch.zeros(L, 10, dtype=torch.int32, device=dev),
                 dense=torch.zeros(2 * V, dtype=torch.int64, device=dev),
                 cand=torch.zeros(V, dtype=torch.int32, device=dev), p=torch.zeros(V, dtype=torch.bfloat16, device=dev),
                 P=torch.zeros(V, dtype=torch.bfloat16, device=dev), is_seed=torch.zeros(V, dtype=torch.uint8, device=dev),
                 scr_a=torch.zeros(2 * nb, dtype=torch.int64, device=dev),    # status words of the two ordered compactions
                 sel=torch.zeros(nb, dtype=torch.int32, device=dev), err=torch.zeros(1, dtype=torch.int32, device=dev),
                 step=torch.zeros(1, dtype=torch.int64, device=dev), seeds0=torch.zeros(S0, dtype=torch.int32, device=dev),
                 seeds_l=[torch.zeros(c["S"], dtype=torch.int32, device=dev) for c in eng.caps],
                 seed_pos=[torch.zeros(c["S"], dtype=torch.int32, device=dev) for c in eng.caps],
                 n_local=torch.zeros(L, dtype=torch.int32, device=dev),
                 counts_host=torch.empty(L * 10, dtype=torch.int32).pin_memory(), rec_host=torch.empty(L, 10, dtype=torch.int32).pin_memory(),
                 nloc_host=torch.empty(L, dtype=torch.int32).pin_memory(), err_host=torch.empty(1, dtype=torch.int32).pin_memory())
        b["slots"] = {0: b}                                       # slot 0 = the entries above; further slots: _slot_bufs
        self._bufs = b
        self.bytes_per_step = L * 2 * V * 8
        return b

    def _slot_bufs(self, slot=0):
        """The per-batch outputs of ``enqueue`` exist once per slot (a pipelined loop holds two batches at a time: the one being
        trained and the one being sampled); the candidate-side scratch is shared."""
        b = self._bufs
        if slot not in b["slots"]:
            dev, L = self.g.device, len(self.nodes_per_layer)
            caps = self.ops.eng.caps
            b["slots"][slot] = dict(counts=torch.zeros(L * 10, dtype=torch.int32, device=dev), rec=torch.zeros(L, 10, dtype=torch.int32, device=dev),
                                    seeds0=torch.zeros_like(b["seeds0"]), n_local=torch.zeros(L, dtype=torch.int32, device=dev),
                                    seeds_l=[torch.zeros(c["S"], dtype=torch.int32, device=dev) for c in caps],
                                    seed_pos=[torch.zeros(c["S"], dtype=torch.int32, device=dev) for c in caps],
                                    counts_host=torch.empty(L * 10, dtype=torch.int32).pin_memory(),
                                    rec_host=torch.empty(L, 10, dtype=torch.int32).pin_memory(), nloc_host=torch.empty(L, dtype=torch.int32).pin_memory())
        return b["slots"][slot]

    def enqueue(self, seeds_g, slot=0, part=None, hook=None, defer=(), layers=None, ready_flag=0, sel_done_flag=0, layer_done_flags=0):
        """One sample_blocks (bandit_sampler.py:341-367) for the global seed list, on the current stream, with capacity-padded
        outputs and NO host sync: safe inside HIP-graph capture.  The step number of the keyed draw lives on the device and
        advances by one per call / replay.  Returns this rank's blocks, input-most first; ``finish()`` reads sizes and errors.

        ``part``: None = everything.  "select" = candidates, draw and kept lists of all layers without the blocks themselves
        (``hook(n)`` is called behind layer n's kept list); "build" = only the blocks (generate_block, :269-339) of a preceding
        "select" with the same slot (``hook(n)`` in front of layer n's block) -- for a loop that builds them on another stream.
        ``defer``: sampling layers whose block this call does NOT build; ``layers``: the only ones a "build" call builds (the loop
        that leaves the input-most block to the backward stream: PipelinedShardedTrainStep; ``ready_flag``: device flag that
        bliss_build_block raises once the blocks' forward arrays are final, BEFORE it sorts the by-source lists of the backward pass;
        ``sel_done_flag``: device flag raised by the last layer's bliss_shard_select_kept: all kept lists are final;
        ``layer_done_flags``: address of L consecutive device flags, flag n raised by layer n's select_kept: its kept list is final.
        ``ready_flag`` applies to the input-most block only).
        Needs one scratch set per layer: a layer's dense maps live until its block is built."""
        if not self.static:
            raise RuntimeError("enqueue() is the static HIP path; construct the sampler without ops")
        g, ops = self.g, self.ops
        eng = ops.eng
        L = len(self.nodes_per_layer)
        order = list(reversed(range(L)))
        fan = [self.nodes_per_layer[b] for b in order]
        S0 = int(seeds_g.numel())
        b = self._ensure_static(S0, fan)
        sb = self._slot_bufs(slot)
        st = torch.cuda.current_stream().cuda_stream
        lib, chk = _lib.lib, _lib.check
        V = eng.V
        select, build = part in (None, "select"), part in (None, "build")
        if (part is not None or defer) and eng.scratch_sets < L:
            raise RuntimeError("split enqueue needs one scratch set per layer (set ops.eng.scratch_sets before the first call)")
        if select:
            if seeds_g.dtype == torch.int32 and seeds_g.is_contiguous() and seeds_g.is_cuda:
                seeds0 = seeds_g                                   # (read by this call's first layer only: no copy into the slot)
            else:
                seeds0 = sb["seeds0"]
                seeds0.copy_(seeds_g.to(torch.int32), non_blocking=True)
        else:
            seeds0 = sb["seeds0"]
        eta_f, ome_f = float(np.float32(self.eta)), float(np.float32(1.0 - self.eta))
        bins = eng._bin_buffers()
        n_touched_ptr = bins["cursor"].data_ptr() + 4 * eng.n_bins
        cur, n_seeds, n_seeds_dev = seeds0, S0, 0
        blocks = []
        for n, layer in enumerate(order):
            cap = eng.caps[n]
            cs, ws = cap["S"], eng.ws[n]
            c_ws, c_out, lay, cnt_ptr, kept_nid = eng._layer_buffers(n, sb["counts"], slot=("dense", slot))
            seeds_l, seed_pos = sb["seeds_l"][n], sb["seed_pos"][n]
            nloc_ptr = sb["n_local"].data_ptr() + 4 * n
            rec_ptr = sb["rec"].data_ptr() + 40 * n
            w_pos = ops.w_pos[layer]
            if select:
                chk(lib.bliss_shard_local_seeds(cur.data_ptr(), n_seeds, n_seeds_dev, g.lo, g.hi, cs, seeds_l.data_ptr(), ws.cand_nid.data_ptr(),
                                                seed_pos.data_ptr(), nloc_ptr, b["err"].data_ptr(), st), "bliss_shard_local_seeds")
                chk(lib.bliss_frontier_prob(C.byref(eng.c_graph), C.byref(eng._set(n)["c_maps"]), w_pos.data_ptr(), seeds_l.data_ptr(), -1, nloc_ptr,
                                            cs, ops.mode | _lib.MODE_PARTIALS, eta_f, ome_f, eng.Eg, C.byref(c_ws), st), "bliss_frontier_prob")
                seed_p2_ptr = ws.seed_acc.data_ptr() + 8 * 4 * cs
                chk(lib.bliss_shard_scatter_partials(seeds_l.data_ptr(), seed_p2_ptr, nloc_ptr, bins["tkey"].data_ptr(), bins["tsum"].data_ptr(),
                                                     n_touched_ptr, b["dense"].data_ptr(), V, b["err"].data_ptr(), st), "bliss_shard_scatter_partials")
                _all_reduce(b["dense"], self.group)                               # THE exchange of this layer (static shape)
                st = torch.cuda.current_stream().cuda_stream
                chk(lib.bliss_shard_candidates(b["dense"].data_ptr(), V, ops.uniform_nodes, b["cand"].data_ptr(), b["p"].data_ptr(),
                                               b["is_seed"].data_ptr(), eng.hist.data_ptr(), rec_ptr, V, b["scr_a"].data_ptr(), b["err"].data_ptr(), st),
                    "bliss_shard_candidates")
                chk(lib.bliss_poisson_scale(eng.hist.data_ptr(), rec_ptr, int(fan[n]), 0.9999, b["sel"].data_ptr(), st), "bliss_poisson_scale")
                chk(lib.bliss_shard_select_kept(b["cand"].data_ptr(), b["p"].data_ptr(), b["is_seed"].data_ptr(), rec_ptr, self.seed, b["step"].data_ptr(),
                                                n, cur.data_ptr(), n_seeds, n_seeds_dev, b["P"].data_ptr(), kept_nid.data_ptr(), c_ws.node_prob,
                                                c_ws.kept_map, cap["K"], V, V, cnt_ptr, nloc_ptr, b["scr_a"].data_ptr(), 1 if n == L - 1 else 0,
                                                (int(layer_done_flags) + 4 * n) if layer_done_flags else (int(sel_done_flag) if n == L - 1 else 0), b["err"].data_ptr(), st),
                    "bliss_shard_select_kept")
                if hook is not None and part == "select":
                    hook(n)
            # (measured and not kept: the block forked to a side stream INSIDE one graph -- correct, but the forked graph replayed at
            # 4.4 ms instead of 1.4: this runtime serialises branches of one graph badly; see PipelinedShardedTrainStep for the form
            # that works: a graph of its own on a third stream, ordered by device flags)
            if build and n not in defer and (layers is None or n in layers):
                if hook is not None and part == "build":
                    hook(n)
                keep_flag, c_ws.block_ready_flag = c_ws.block_ready_flag, ((ready_flag if n == L - 1 else 0) or c_ws.block_ready_flag)
                chk(lib.bliss_build_block(C.byref(eng.c_graph), C.byref(eng._set(n)["c_maps"]), w_pos.data_ptr(), seeds_l.data_ptr(), cs, ops.mode,
                                          eta_f, ome_f, eng.Eg, C.byref(c_ws), C.byref(c_out), st), "bliss_build_block")
                c_ws.block_ready_flag = keep_flag
            b_indptr, b_src, b_dst, b_pos, b_eid, b_w, b_q, kept, node_prob, cdev, t_indptr, t_edge = lay
            if not select:                                         # (the block objects exist: made by the "select" part)
                cur, n_seeds, n_seeds_dev = kept, -1, cnt_ptr + 12
                continue
            # (the destinations' ids = this rank's seeds: bliss_shard_local_seeds wrote them, padded with the list's first entry, into
            # a persistent per-slot buffer -- block objects of one slot are interchangeable between graphs)
            blk = ShardBlock(g, cap["K"], cs, b_indptr, b_src, b_dst, b_pos, b_eid, kept, seed_pos, dst_nid=seeds_l)
            blk._edge_weights, blk._q, blk._node_prob = b_w, b_q, node_prob
            blk._counts, blk._counts_dev, blk._layer = None, cdev, layer
            blk._nnz_ptr = sb["counts"].data_ptr() + 40 * n + 16
            blk._xcap = int(cap["B"])
            if t_indptr is not None:
                blk._transposed = (t_indptr, t_edge)
            blk.edata["edge_weights"], blk.edata["q_ij"], blk.srcdata["node_prob"] = b_w, b_q, node_prob
            blocks.insert(0, blk)
            cur, n_seeds, n_seeds_dev = kept, -1, cnt_ptr + 12
        if not select:
            return None
        self._static_blocks = blocks                             # (the step number went up by one in the last layer's select_kept)
        return blocks

    def finish(self, slot=0):
        """After a synchronisation: true sizes per layer (sampling order) and the error check."""
        b, eb = self._slot_bufs(slot), self._bufs
        b["counts_host"].copy_(b["counts"]); b["rec_host"].copy_(b["rec"]); b["nloc_host"].copy_(b["n_local"]); eb["err_host"].copy_(eb["err"])
        torch.cuda.current_stream().synchronize()
        raw = b["counts_host"].numpy().tobytes()
        L = len(self.nodes_per_layer)
        cnts = [_lib.LayerCounts.from_buffer_copy(raw[40 * n: 40 * n + 40]) for n in range(L)]
        bad = int(eb["err_host"][0])
        for c in cnts:
            bad |= c.err
        if bad:
            raise RuntimeError(f"static sharded sampler kernel error 0x{bad:x}: {_lib.err_string(bad)}")
        recs = [_lib.LayerCounts.from_buffer_copy(b["rec_host"][n].numpy().tobytes()) for n in range(L)]
        self.trace = [dict(C=r.C, scale=(float(r.c), bool(r.all_one), int(r.iters))) for r in recs]
        return [dict(S=int(b["nloc_host"][n]), E=c.E, K=c.K, B=c.B) for n, c in enumerate(cnts)]

    def finish_nothrow(self, slot=0):
        """finish() for diagnostics: sizes and the per-layer error words, no exception."""
        b = self._slot_bufs(slot)
        b["counts_host"].copy_(b["counts"]); b["nloc_host"].copy_(b["n_local"])
        torch.cuda.current_stream().synchronize()
        raw = b["counts_host"].numpy().tobytes()
        cnts = [_lib.LayerCounts.from_buffer_copy(raw[40 * n: 40 * n + 40]) for n in range(len(self.nodes_per_layer))]
        return [dict(S=int(b["nloc_host"][n]), E=c.E, K=c.K, B=c.B, err=c.err) for n, c in enumerate(cnts)]

    def sample_blocks(self, seeds_g, step=None):
        if not self.static:
            return self.sample_blocks_generic(seeds_g, step)
        if step is not None:
            self._ensure_static(int(seeds_g.numel()), [self.nodes_per_layer[b] for b in reversed(range(len(self.nodes_per_layer)))])
            self._bufs["step"].fill_(int(step))
        blocks = self.enqueue(seeds_g)
        self.sizes = self.finish()
        return blocks[0].srcdata[NID], blocks[-1].dstdata[NID], blocks

    # ---------------------------------------------------------------- EXP3 (owner-local update, global norm)
    def exp3(self, mfgs, g=None):
        """bandit_sampler.py:251-267 on the owned in-edges of every block, then the global L1 renormalisation."""
        ops = self.ops
        if hasattr(ops, "exp3_update_all") and 0 < len(mfgs) <= _lib.EXP3_MAX_BLOCKS and all(m.num_edges() > 0 for m in mfgs):
            ops.exp3_update_all(mfgs, self._delta_f)             # one launch for all blocks
        else:
            for mfg in mfgs:
                ops.exp3_update(mfg, mfg.srcdata["embed_norm"], self._delta_f)
        if not hasattr(ops, "row_sum"):                          # (test doubles: one normalisation per layer)
            for idx in range(len(mfgs)):
                ops.normalize(idx, self.group)
            return
        # the layers' rows are disjoint: ONE all-reduce of all exact row sums, then the normalisations
        L = len(mfgs)
        limbs = _all_reduce(ops.row_sum[:L].clone(), self.group)
        st = torch.cuda.current_stream().cuda_stream
        if L <= _lib.EXP3_MAX_BLOCKS and limbs.is_contiguous():   # all rows in ONE launch
            rows = (_lib.Exp3Block * L)()
            for idx in range(L):
                rows[idx] = _lib.Exp3Block(ops.w_pos[idx].data_ptr(), ops.row_sum[idx].data_ptr(), ops.scratch[idx].data_ptr(),
                                           ops.norms[idx:].data_ptr(), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
            _lib.check(_lib.lib.bliss_exp3_normalize_global_rows(rows, L, self.g.num_edges(), limbs.data_ptr(), limbs.stride(0), st),
                       "bliss_exp3_normalize_global_rows")
            return
        for idx in range(L):
            _lib.check(_lib.lib.bliss_exp3_normalize_global(ops.w_pos[idx].data_ptr(), self.g.num_edges(), ops.row_sum[idx].data_ptr(),
                                                            limbs[idx].data_ptr(), ops.scratch[idx].data_ptr(), ops.norms[idx:].data_ptr(), st),
                       "bliss_exp3_normalize_global")

    def check_errors(self):
        if hasattr(self.ops, "check_errors"):
            self.ops.check_errors()
        if self._bufs is not None:
            bits = int(self._bufs["err"].item())
            if bits:
                raise RuntimeError(f"static sharded sampler kernel error 0x{bits:x}: {_lib.err_string(bits)}")


# ------------------------------------------------------------------------------------------- the model step
def _reduce_rows_(out, group):
    """In-place sum over the ranks of a zero-padded bf16 row buffer, as int32 words (exact: one non-zero contributor per word)."""
    if (out.shape[1] * out.element_size()) % 4 == 0 and out.is_contiguous():
        _all_reduce(out.view(torch.int32), group)
        return out
    f = out.float()                                              # odd row length: fp32 carries bf16 exactly
    _all_reduce(f, group)
    return f.to(out.dtype)


class _PlaceAndReduce(torch.autograd.Function):
    """_PlaceRows + the halo all-reduce in one node (no copy in between): out = sum over ranks of (zeros; out[idx[i]] = h[i])."""

    @staticmethod
    def forward(ctx, h, idx, n_rows, group, group_bwd=None):
        ctx.save_for_backward(idx)
        ctx.n_rows, ctx.group = int(n_rows), (group if group_bwd is None else group_bwd)
        out = _zeros((ctx.n_rows + 1, h.shape[1]), h.dtype, h.device)
        out.index_copy_(0, idx, h)
        return _reduce_rows_(out, group)[:ctx.n_rows]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gp = torch.empty((ctx.n_rows + 1,) + tuple(g.shape[1:]), dtype=torch.float32, device=g.device)
        gp[:ctx.n_rows] = g
        gp[ctx.n_rows:].fill_(0)
        _all_reduce(gp, ctx.group)                               # a row's gradient: the sum over the ranks that consumed it
        return torch.index_select(gp, 0, idx).to(g.dtype), None, None, None, None


class _HaloAllReduce(torch.autograd.Function):
    """Block inputs over all ranks: every rank wrote the rows it owns into a zero buffer of the block's capacity; the sum of
    the buffers is the full input.  Sent as int32 words (two bf16 each: x + 0 + .. + 0 is exact and an integer is a type both
    RCCL and gloo reduce).  Backward: the gradient of a row goes back to the rank that produced it, summed over all consumers
    -- the all-reduce of the fp32 gradient buffer, of which autograd then takes this rank's rows."""

    @staticmethod
    def forward(ctx, buf, group):
        ctx.group = group
        out = buf.contiguous().clone()
        if (out.shape[1] * out.element_size()) % 4 == 0:
            _all_reduce(out.view(torch.int32), group)
        else:                                                    # odd row length: fp32 carries bf16 exactly
            f = out.float()
            _all_reduce(f, group)
            out = f.to(buf.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        gf = g.float().contiguous()
        _all_reduce(gf, ctx.group)
        return gf.to(g.dtype), None


# Zero-filled buffers inside the (captured) step come from a FILL KERNEL, never from torch.zeros / zero_(): those are
# hipMemsetAsync, and memset nodes of a captured HIP graph were observed to leave their buffer stale on replay (ROCm 7.2; the
# same finding as csrc/sampler.hip:k_seg_scan and csrc/shard_dense.hip:k_sd_zero).  That includes the zero buffers autograd
# creates on its own -- the backward of x[idx] / index_select / gather / nll_loss is "zeros, then scatter-add": replayed with a
# stale buffer the gradients of earlier steps pile up and the parameters overflow within a few steps (seen on the Reddit-like
# step with calibrated capacities).  Hence the three small autograd Functions below and the one-hot cross-entropy in _body.
def _zeros(shape, dtype, device):
    return torch.empty(shape, dtype=dtype, device=device).fill_(0)


class _TakeRows(torch.autograd.Function):
    """x[idx] (idx may repeat); backward: fill-kernel zeros + index_add_."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return torch.index_select(x, 0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gx = _zeros((ctx.n,) + tuple(g.shape[1:]), g.dtype, g.device)
        gx.index_add_(0, idx, g.contiguous())
        return gx, None


class _PlaceRows(torch.autograd.Function):
    """out = zeros[n_rows]; out[idx[i]] = h[i], rows with idx == n_rows (the sink) dropped.  Backward: g[idx] (the sink: 0)."""

    @staticmethod
    def forward(ctx, h, idx, n_rows):
        ctx.save_for_backward(idx)
        ctx.n_rows = int(n_rows)
        out = _zeros((ctx.n_rows + 1, h.shape[1]), h.dtype, h.device)
        out.index_copy_(0, idx, h)
        return out[:ctx.n_rows]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gp = torch.empty((ctx.n_rows + 1,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        gp[:ctx.n_rows] = g
        gp[ctx.n_rows:].fill_(0)
        return torch.index_select(gp, 0, idx), None, None


def _hip_rows(x, pos):
    """The in-tree row kernels (csrc/shard_dense.hip) take bf16 rows of even length on the GPU and int32 positions."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] % 2 == 0 and pos.dtype == torch.int32 and pos.is_contiguous())


def _place_rows(src, pos, n_dev, n_rows):
    src = src.contiguous()
    out = torch.empty(int(n_rows), src.shape[1], dtype=src.dtype, device=src.device)
    _lib.check(_lib.lib.bliss_shard_place_rows(src.data_ptr(), src.stride(0), pos.data_ptr(), n_dev.data_ptr(), min(src.shape[0], pos.numel()),
                                               out.data_ptr(), out.stride(0), int(n_rows), src.shape[1],
                                               torch.cuda.current_stream().cuda_stream), "bliss_shard_place_rows")
    return out


def _take_rows(src, pos, n_dev, cap_s):
    src = src.contiguous()
    out = torch.empty(int(cap_s), src.shape[1], dtype=torch.bfloat16, device=src.device)
    _lib.check(_lib.lib.bliss_shard_take_rows(src.data_ptr(), 1 if src.dtype == torch.float32 else 0, src.stride(0), src.shape[0], pos.data_ptr(),
                                              n_dev.data_ptr(), int(cap_s), out.data_ptr(), out.stride(0), src.shape[1],
                                              torch.cuda.current_stream().cuda_stream), "bliss_shard_take_rows")
    return out


class _TakeRowsHip(torch.autograd.Function):
    """_TakeRows in one launch each way: x[pos[j]] for this rank's j < *n_dev rows (+0 behind); backward: the rows back at their
    positions, +0 elsewhere (pos ascending: no zero-fill, no atomics)."""

    @staticmethod
    def forward(ctx, x, pos, n_dev):
        ctx.save_for_backward(pos, n_dev)
        ctx.n = x.shape[0]
        return _take_rows(x, pos, n_dev, pos.numel())

    @staticmethod
    def backward(ctx, g):
        pos, n_dev = ctx.saved_tensors
        return _place_rows(g, pos, n_dev, ctx.n), None, None


class _PlaceAndReduceHip(torch.autograd.Function):
    """_PlaceAndReduce on the in-tree row kernels: the zero-padded buffer in one launch, the halo all-reduce on it in place;
    backward: the fp32 gradient buffer summed over the ranks, then this rank's rows of it, rounded to bf16, in one launch."""

    @staticmethod
    def forward(ctx, h, pos, n_dev, n_rows, group, group_bwd=None):
        ctx.save_for_backward(pos, n_dev)
        ctx.cap_s, ctx.group = h.shape[0], (group if group_bwd is None else group_bwd)
        return _reduce_rows_(_place_rows(h, pos, n_dev, n_rows), group)

    @staticmethod
    def backward(ctx, g):
        pos, n_dev = ctx.saved_tensors
        gp = g.float().contiguous()
        _all_reduce(gp, ctx.group)                               # a row's gradient: the sum over the ranks that consumed it
        return _take_rows(gp, pos, n_dev, ctx.cap_s), None, None, None, None, None


def _xdev_of(t, group=None):
    return t.device if dist.get_backend(group) == "nccl" else torch.device("cpu")


def halo_all_reduce(buf, group=None):
    return _HaloAllReduce.apply(buf, group)


def measure_caps(shard, nodes_per_layer, model, batch, loader, steps=4, eta=0.4, seed=0, multilabel=False, group=None,
                 k_margin=1.4, b_margin=2.0):
    """Static capacities for ``DenseShardedSampler(fixed_caps=...)`` from ``steps`` steps of a THROW-AWAY copy of the model and
    sampler state: the run that is measured (and possibly captured) afterwards starts from untouched parameters and EXP3 rows,
    and its first launches happen inside ``capture``'s warm-up -- see StaticShardedTrainStep.calibrate for the in-place form."""
    import copy
    tmp_sampler = DenseShardedSampler(shard, nodes_per_layer, eta=eta, seed=seed, group=group)
    tmp_step = StaticShardedTrainStep(shard, tmp_sampler, copy.deepcopy(model), batch, multilabel=multilabel, group=group)
    tmp_step.calibrate(loader, steps=steps, k_margin=k_margin, b_margin=b_margin)
    caps = tmp_sampler.ops.fixed_caps
    tmp_step.close()
    del tmp_step, tmp_sampler
    import gc
    gc.collect()
    torch.cuda.synchronize()
    return caps


class StaticShardedTrainStep:
    """One optimiser step of ModelLightning (train_lightning.py:100-168, 205-216, 463-471) over destination-range shards with
    static shapes: every rank contributes ``batch`` seeds it owns (the global batch is their concatenation, rank by rank), the
    sampler is ``DenseShardedSampler.enqueue``, block inputs travel as capacity-sized all-reduces, the loss is the mean over the
    GLOBAL batch.  No size reaches the host inside ``_body``: with RCCL the whole step records into ONE HIP graph (``capture``).

    Collectives per step: 1 all-gather (seeds) + L dense all-reduces (sampler) + L halo all-reduces (+ L - 1 backward) + 1
    (gradients) + L (EXP3 row sums) + 1 (loss): all of fixed shape; ``bytes_per_step`` adds up what a rank contributes."""

    def __init__(self, shard, sampler, model, batch, lr=0.002, multilabel=False, group=None):
        import torch.nn as nn
        from .train import make_adam
        if not sampler.static:
            raise ValueError("StaticShardedTrainStep needs a DenseShardedSampler on its static HIP path (ops=None)")
        self.g, self.sampler, self.model, self.group, self.batch = shard, sampler, model, group, int(batch)
        self.multilabel = bool(multilabel)
        self.opt = make_adam(model, lr, capturable=True)
        dev = shard.device
        self.my_seeds = torch.zeros(self.batch, dtype=torch.int32, device=dev)
        self.seeds_g = torch.zeros(self.batch * shard.world, dtype=torch.int32, device=dev) if shard.world > 1 else self.my_seeds
        self.loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ce_state = torch.zeros(2, dtype=torch.int32, device=dev)                           # the loss kernel's [0] ticket, [1] error word
        self.graph = None
        self.last = {}
        self.bytes_per_step = 0
        self._aranges = {}

    def _group_bwd(self):
        """The communicator of the collectives the BACKWARD pass issues (None: the forward's).  A loop that runs the backward
        pass on a second stream beside the sampler gives it a communicator of its own: one RCCL communicator must see its
        collectives in one order on every rank, which two streams do not guarantee."""
        return getattr(self, "group_b", None)

    def _arange(self, n):
        """0 .. n-1 (int64) on the device, created once per length: five launches less inside the step."""
        t = self._aranges.get(int(n))
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                return torch.arange(int(n), device=self.g.device)
            t = self._aranges[int(n)] = torch.arange(int(n), device=self.g.device)
        return t

    def _gather_seeds(self):
        if self.seeds_g is self.my_seeds:                         # (a world of one rank: the global list IS this rank's)
            return
        if self.g.world == 1:
            self.seeds_g.copy_(self.my_seeds)
        elif dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(self.seeds_g, self.my_seeds, group=self.group)
        else:
            parts = [torch.empty(self.batch, dtype=torch.int32) for _ in range(self.g.world)]
            dist.all_gather(parts, self.my_seeds.cpu(), group=self.group)
            self.seeds_g.copy_(torch.cat(parts))

    def _fused_layer_ok(self, layer, h):
        """The MFMA tail applies to an aggregate-first SAGEConv (in <= out) of the reference's kind: bf16 on the GPU, ReLU between
        the layers, plain dropout, bias on fc_self, within the tile kernel's limits.  BLISS_SHARD_MFMA=0: library GEMMs."""
        import os
        from .nn import tile_gemm_ok
        m = self.model
        act = getattr(m, "activation", None)
        relu = act in (torch.relu, torch.nn.functional.relu) or isinstance(act, torch.nn.ReLU)
        return (os.environ.get("BLISS_SHARD_MFMA", "1") != "0" and relu and hasattr(m, "_dropout_state") and isinstance(m.dropout, torch.nn.Dropout)
                and h.is_cuda and h.dtype == torch.bfloat16 and layer._in_src_feats <= layer._out_feats
                and tile_gemm_ok(layer._in_src_feats, layer._out_feats) and layer.fc_self.bias is not None and layer.norm is None
                and layer.activation is None and layer.feat_drop.p == 0)

    def _split_layer_ok(self, layer, h):
        """The same for a Linear-first SAGEConv (in > out): nn._SageLinearSplit + the fused epilogue."""
        import os
        from .nn import tile_gemm_ok
        m = self.model
        act = getattr(m, "activation", None)
        relu = act in (torch.relu, torch.nn.functional.relu) or isinstance(act, torch.nn.ReLU)
        return (os.environ.get("BLISS_SHARD_MFMA", "1") != "0" and relu and hasattr(m, "_dropout_state") and isinstance(m.dropout, torch.nn.Dropout)
                and h.is_cuda and h.dtype == torch.bfloat16 and layer._in_src_feats > layer._out_feats
                and tile_gemm_ok(layer._in_src_feats, layer._out_feats) and layer.fc_self.bias is not None and layer.norm is None
                and layer.activation is None and layer.feat_drop.p == 0)

    def _forward(self, blocks, slot=0, defer_output=False):
        """SAGE.forward (model.py:312-333) over this rank's blocks.  ``defer_output``: stop in front of the output layer's compute
        -- its input rows are in place and their norms stored (all the EXP3 update reads, section 6 item 16) -- and return a
        pending record for ``_forward_output`` (the loop that runs the output layer on the backward stream)."""
        from .nn import embed_norm
        g, model, grp = self.g, self.model, self.group
        lo, hi = g.lo, g.hi
        n_own = hi - lo
        L = len(blocks)
        n_local = self.sampler._slot_bufs(slot)["n_local"]
        h, halo_bytes = None, 0
        for l, (layer, blk) in enumerate(zip(model.layers, blocks)):
            cap_k = blk.num_src_nodes()
            if l == 0:                                             # train_lightning.py:138, owner side: my feature rows, zeros elsewhere
                nid = blk.srcdata[NID]
                feats = g.ndata_owned["features"]
                if (feats.dtype == torch.bfloat16 and feats.is_cuda and feats.stride(1) == 1 and feats.shape[1] % 2 == 0 and feats.stride(0) % 2 == 0
                        and nid.dtype == torch.int32 and nid.is_contiguous()):
                    buf = torch.empty(cap_k, feats.shape[1], dtype=feats.dtype, device=feats.device)          # one launch: csrc/shard_dense.hip
                    _lib.check(_lib.lib.bliss_shard_pack_rows(nid.data_ptr(), blk._counts_dev.data_ptr() + 12, cap_k, lo, hi, feats.data_ptr(),
                                                              feats.stride(0), feats.shape[1], buf.data_ptr(), buf.stride(0),
                                                              torch.cuda.current_stream().cuda_stream), "bliss_shard_pack_rows")
                else:
                    valid = (self._arange(cap_k) < blk._counts_dev[3]) & (nid >= lo) & (nid < hi)
                    rows = feats[(nid.long() - lo).clamp(0, n_own - 1)]
                    buf = torch.where(valid[:, None], rows, 0.0)      # (+0 bits: x * 0 can be -0, and the words are summed as integers)
                h_src = _reduce_rows_(buf, grp)                    # (fresh, no gradient: reduced in place)
            else:                                                  # the rows I computed, at their positions of this block's source list
                prev = blocks[l - 1]
                cap_s = prev.num_dst_nodes()
                if _hip_rows(h, prev.dst_pos):                      # block l-1 <-> sampling layer L-l
                    h_src = _PlaceAndReduceHip.apply(h, prev.dst_pos, n_local[L - l: L - l + 1], cap_k, grp, self._group_bwd())
                else:
                    idx = torch.where(self._arange(cap_s) < n_local[L - l], prev.dst_pos.long(), cap_k)
                    h_src = _PlaceAndReduce.apply(h, idx, cap_k, grp, self._group_bwd())
            halo_bytes += cap_k * h_src.shape[1] * h_src.element_size() * (1 if l == 0 else 3)      # (+ the fp32 gradient buffer)
            if defer_output and l == L - 1:
                blk.srcdata["embed_norm"] = embed_norm(h_src)      # model.py:318-320 (a Linear-first launch would leave the same bits)
                self._halo_bytes = halo_bytes
                return ("pending", h_src)
            h = self._layer_compute(l, L, layer, blk, h_src, n_local, True)
        self._halo_bytes = halo_bytes
        return h

    def _forward_output(self, blocks, pending, slot=0):
        """The output layer of a ``_forward(..., defer_output=True)``."""
        L = len(blocks)
        return self._layer_compute(L - 1, L, self.model.layers[L - 1], blocks[L - 1], pending[1],
                                   self.sampler._slot_bufs(slot)["n_local"], False)

    def _layer_compute(self, l, L, layer, blk, h_src, n_local, set_norm):
        from .nn import embed_norm
        model = self.model
        split = self._split_layer_ok(layer, h_src)
        if not split and set_norm:
            blk.srcdata["embed_norm"] = embed_norm(h_src)      # model.py:318-320
        # (the padding entries of dst_pos all point at row 0: advanced indexing's backward would sort and serialise them --
        # 0.94 ms per layer on the Reddit-like step; index_add_ is atomic, and the duplicates carry zero gradients)
        if split and blk.dst_pos.dtype == torch.int32 and blk.dst_pos.is_contiguous():
            h_dst = None                                       # (the Linear-first launch gathers its destination rows itself)
        elif _hip_rows(h_src, blk.dst_pos):
            h_dst = _TakeRowsHip.apply(h_src, blk.dst_pos, n_local[L - 1 - l: L - l])
        else:
            h_dst = _TakeRows.apply(h_src, blk.dst_pos.long())
        if self._fused_layer_ok(layer, h_src):
            # an aggregate-first layer's tail on the in-tree MFMA tiles (nn._SageDualLinear: fc_neigh(h_neigh) + fc_self(h_dst) +
            # bias, ReLU and dropout in ONE launch, its backward on csrc/sage_bwd.hip) -- the true row count from the device
            from .nn import _SageDualLinear, weighted_aggregate
            last = l == L - 1
            p = model.dropout.p if (model.training and not last) else 0.0
            ctr, seed = model._dropout_state(l, h_src.device) if p > 0 else (None, 0)
            agg = weighted_aggregate(blk, h_src, blk.edata["edge_weights"], mean=True)
            rows_dev = n_local.data_ptr() + 4 * (L - 1 - l)
            h, _ = _SageDualLinear.apply(agg, h_dst, layer.fc_neigh.weight, layer.fc_self.weight, layer.fc_self.bias, not last, p, ctr, seed,
                                         blk.num_dst_nodes(), rows_dev)
        elif split:
            # a Linear-first layer: fc_neigh over the source rows, fc_self + bias over the destination rows and the source rows'
            # norms (:318-320, same bits as embed_norm) in ONE launch on the MFMA tiles; then the aggregation; then the sum, ReLU
            # and dropout (:321-333) in one launch.  True row counts from the device (K of the block, this rank's destinations)
            from .nn import _SageLinearSplit, sage_epilogue, weighted_aggregate
            last = l == L - 1
            p = model.dropout.p if (model.training and not last) else 0.0
            ctr, seed = model._dropout_state(l, h_src.device) if p > 0 else (None, 0)
            z, y, in_norm = _SageLinearSplit.apply(h_src, h_dst, layer.fc_neigh.weight, layer.fc_self.weight, layer.fc_self.bias,
                                                   blk._counts_dev.data_ptr() + 12, n_local.data_ptr() + 4 * (L - 1 - l),
                                                   blk.dst_pos if h_dst is None else None)
            if set_norm:
                blk.srcdata["embed_norm"] = in_norm
            agg = weighted_aggregate(blk, z, blk.edata["edge_weights"], mean=True)
            if last and self._fused_loss_ok(y):
                h = (y, agg)                                   # (the loss launch adds them: bliss_cross_entropy_masked)
            else:
                h = (y + agg) if last else sage_epilogue(y, agg, p, ctr, seed)[0]
        else:
            h = layer(blk, (h_src, h_dst), edge_weight=blk.edata["edge_weights"])
            if l < L - 1:
                h = model.dropout(model.activation(h))         # :330-332
        return h

    def _fused_loss_ok(self, logits):
        """nn.CrossEntropyLoss on the in-tree kernel (csrc/loss.hip, masked form): bf16 logits on the GPU, class-index labels."""
        import os
        lab = self.g.ndata_owned.get("labels") if hasattr(self.g.ndata_owned, "get") else None
        return (os.environ.get("BLISS_SHARD_FUSED_LOSS", "1") != "0" and not self.multilabel and logits.is_cuda
                and logits.dtype == torch.bfloat16 and logits.dim() == 2 and logits.stride(1) == 1 and lab is not None
                and lab.dtype == torch.int64 and lab.dim() == 1 and lab.is_contiguous())

    def _fused_loss_backward_step(self, blocks, pred, slot, grp):
        """The loss and its gradient in ONE launch (masked mean cross-entropy over this rank's output seeds, divided by the GLOBAL
        batch, the output layer's two addends summed inside), ``backward`` from that gradient, gradient all-reduce, Adam."""
        g = self.g
        a, b = pred if isinstance(pred, tuple) else (pred, None)
        last = blocks[-1]
        nid = last.dstdata[NID]
        nid = nid if (nid.dtype == torch.int32 and nid.is_contiguous()) else nid.to(torch.int32).contiguous()
        lab = g.ndata_owned["labels"]
        n_rows, n_cls = a.shape
        n_mine = self.sampler._slot_bufs(slot)["n_local"]                                         # [0]: sampling layer 0 = the output block
        dx = torch.empty(n_rows, n_cls, dtype=torch.bfloat16, device=a.device)
        rows = torch.empty(n_rows, dtype=torch.float32, device=a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        ad, bd = a.detach(), (None if b is None else b.detach())
        _lib.check(_lib.lib.bliss_cross_entropy_masked(ad.data_ptr(), ad.stride(0), 0 if bd is None else bd.data_ptr(),
                                                       0 if bd is None else bd.stride(0), lab.data_ptr(), lab.numel(), nid.data_ptr(), g.lo,
                                                       n_rows, n_mine.data_ptr(), float(self.batch * g.world), n_cls, rows.data_ptr(),
                                                       dx.data_ptr(), dx.stride(0), loss.data_ptr(), self._ce_state.data_ptr(),
                                                       self._ce_state.data_ptr() + 4, torch.cuda.current_stream().cuda_stream),
                   "bliss_cross_entropy_masked")
        self.opt.zero_grad(set_to_none=True)
        if b is None:
            a.backward(dx)
        else:
            torch.autograd.backward([a, b], [dx, dx])
        self._allreduce_gradients(1.0, grp)                      # (the gradients carry 1 / global batch already)
        self.opt.step()
        _all_reduce(loss, grp)
        self.loss_dev.copy_(loss)
        n_par = sum(p.numel() for p in self.model.parameters())
        self.bytes_per_step = self.sampler.bytes_per_step + self._halo_bytes + 4 * n_par + 4 * self.batch + 96 * 8 * len(blocks) + 4
        return ad if bd is None else ad + bd

    def _loss_backward_step(self, blocks, pred, slot=0):
        """Masked loss over this rank's output seeds, backward, gradient all-reduce, Adam, the global mean loss (the collectives of
        this part go through ``_group_bwd()`` when the loop runs it beside the sampler)."""
        g = self.g
        grp = self._group_bwd() if self._group_bwd() is not None else self.group
        if isinstance(pred, tuple) or self._fused_loss_ok(pred):
            return self._fused_loss_backward_step(blocks, pred, slot, grp)
        last = blocks[-1]
        cap_s = last.num_dst_nodes()
        n_mine = self.sampler._slot_bufs(slot)["n_local"][0]                                      # (sampling layer 0 = the output block)
        mask = self._arange(cap_s) < n_mine
        pred = torch.where(mask[:, None], pred, 0.0)                                              # (rows beyond the count: whatever the layer left)
        y = g.ndata_owned["labels"][(last.dstdata[NID].long() - g.lo).clamp(0, g.hi - g.lo - 1)]  # :139
        n_global = self.batch * g.world
        if self.multilabel:
            per_row = torch.nn.functional.binary_cross_entropy_with_logits(pred.float(), y.float(), reduction="none").sum(1)
            scale = 1.0 / (n_global * pred.shape[1])
        else:                                                  # cross-entropy through a one-hot product (nll_loss's backward is zeros + scatter)
            logp = torch.log_softmax(pred.float(), 1)
            onehot = self._arange(pred.shape[1])[None, :] == y[:, None]
            per_row = -(logp * onehot).sum(1)
            scale = 1.0 / n_global
        loss_sum = (per_row * mask).sum()                                                         # padding rows: no loss, no gradient
        self.opt.zero_grad(set_to_none=True)
        loss_sum.backward()
        self._allreduce_gradients(scale, grp)
        self.opt.step()
        tot = loss_sum.detach().float().reshape(1).clone()
        _all_reduce(tot, grp)
        self.loss_dev.copy_(tot * scale)
        n_par = sum(p.numel() for p in self.model.parameters())
        self.bytes_per_step = self.sampler.bytes_per_step + self._halo_bytes + 4 * n_par + 4 * self.batch + 96 * 8 * len(blocks) + 4
        return pred.detach()

    def _body(self):
        self._gather_seeds()
        blocks = self.sampler.enqueue(self.seeds_g)
        pred = self._forward(blocks)                                                              # train_lightning.py:138-141
        pred = self._loss_backward_step(blocks, pred)
        self.sampler.exp3(blocks)                                                                 # :469-471
        self.last = dict(mfgs=blocks, pred=pred)

    def _allreduce_gradients(self, scale, group=None):
        """Sum the gradients over the ranks and scale by 1 / global batch: ONE flat fp32 bucket (cat, cast, all-reduce, scale,
        cast, one multi-tensor copy back) instead of a cast and a copy per parameter."""
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        if not grads or (self.g.world == 1 and scale == 1.0):    # (alone and already scaled: the gradients are final)
            return
        flat = torch.cat([g_.reshape(-1) for g_ in grads]).float()
        _all_reduce(flat, self.group if group is None else group)
        if scale != 1.0:
            flat = flat.mul_(scale)
        flat = flat.to(grads[0].dtype)
        views, off = [], 0
        for g_ in grads:
            n = g_.numel()
            views.append(flat[off:off + n].view_as(g_))
            off += n
        torch._foreach_copy_(grads, views)

    def __call__(self, my_seeds):
        """``my_seeds``: the ``batch`` seeds this rank contributes (ids it owns).  Returns the global mean loss (a device scalar
        read after the step's one synchronisation)."""
        self.my_seeds.copy_(my_seeds.to(torch.int32), non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()
        return self.loss_dev

    def calibrate(self, loader, steps=4, k_margin=1.4, b_margin=2.0):
        """Static capacities from observed sizes instead of the worst-case formula of shard._HipShardOps.set_caps (2 (fanout + S)
        kept nodes per layer): ``steps`` eager static steps, the largest K and B per layer over all ranks (all-reduce MAX), times
        a margin.  Halo buffers, padded GEMM rows and the bytes that cross xGMI shrink with them; a later step that exceeds a
        capacity raises the kernels' capacity error bits (``finish`` / ``check_errors``)."""
        L = len(self.sampler.nodes_per_layer)
        mx = torch.zeros(L, 2, dtype=torch.int64, device=self.g.device)
        for _ in range(steps):
            self(next(loader))
            _, sizes = self.finish()
            mx = torch.maximum(mx, torch.tensor([[z["K"], z["B"]] for z in sizes], dtype=torch.int64, device=mx.device))
        if self.g.world > 1:
            x = mx.to(_xdev_of(mx, self.group))
            dist.all_reduce(x, op=dist.ReduceOp.MAX, group=self.group)
            mx = x.to(mx.device)
        up = lambda x, m: (int(x) + m - 1) // m * m
        self.sampler.ops.fixed_caps = [dict(K=up(k_margin * int(k) + 256, 64), B=up(b_margin * int(b) + 4096, 1024)) for k, b in mx.tolist()]
        self.sampler._bufs = None                                  # re-sized on the next enqueue

    def capture(self, loader, warmup=2):
        """Warm-up steps on a side stream, then record the step -- sampler, collectives, model, Adam, EXP3 -- into one HIP graph
        (RCCL only: gloo's collectives run on the host)."""
        if self.g.world > 1 and dist.get_backend(self.group) != "nccl":
            raise RuntimeError("graph capture needs the collectives on the device (backend nccl)")
        import gc
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self(next(loader))
            torch.cuda.current_stream().synchronize()
        torch.cuda.current_stream().wait_stream(side)
        self.sampler.check_errors()
        self.last = {}
        gc.collect()
        torch.cuda.synchronize()
        self.my_seeds.copy_(next(loader).to(torch.int32))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, **_cap_kw()):
            self._body()
        self.graph = graph
        self.graph.replay()                                        # the capture executed nothing: this batch is a real step
        torch.cuda.synchronize()

    def finish(self):
        """The step's one synchronisation: true block sizes, error words, the loss."""
        sizes = self.sampler.finish()
        self.sampler.check_errors()
        self._check_loss_errors()
        return float(self.loss_dev.item()), sizes

    def _check_loss_errors(self):
        if int(self._ce_state[1].item()):
            raise RuntimeError("the loss kernel met a label outside [0, n_classes) or a destination id outside this rank's range")

    def close(self):
        import gc
        torch.cuda.synchronize()
        self.graph, self.last = None, {}
        gc.collect()
        torch.cuda.synchronize()



class PipelinedShardedTrainStep(StaticShardedTrainStep):
    """The static sharded step as a two-stream loop (the single-GPU loop's shape, train.PipelinedTrainStep): on the critical stream
    F(t) -> X(t) -> S(t+1) -- the forward pass of the batch sampled one call earlier, its EXP3 update, the next batch's sampling --
    and beside it, on a second stream, loss / backward / gradient all-reduce / Adam of batch t, which F(t+1) waits for.  Same
    arithmetic, same order of the data flow as ``StaticShardedTrainStep`` (X(t) reads the norms F(t) stored; it no longer waits for
    Adam): the two train identically.

    (Measured and not kept: the next batch's blocks built on the backward stream behind B -- 1.17 ms/step instead of 1.10: B + the
    block builds then form the longer chain.)

    Two block slots (the batch being trained, the batch being sampled).  The backward stream's collectives -- halo gradients,
    parameter gradients, the loss -- go through ``group_b``, a communicator of their own: one RCCL communicator must see its
    collectives in one order on every rank, and two streams do not guarantee one.  Graph mode (``capture``): per slot three graphs
    -- F + X, S, B -- replayed from the two streams and ordered by two events; forks INSIDE one graph serialise on this runtime."""

    def __init__(self, shard, sampler, model, batch, lr=0.002, multilabel=False, group=None, group_b=None):
        super().__init__(shard, sampler, model, batch, lr=lr, multilabel=multilabel, group=group)
        if shard.world > 1 and group_b is None:
            group_b = dist.new_group(backend=dist.get_backend(group))
        self.group_b = group_b
        self.side, self.third = torch.cuda.Stream(), torch.cuda.Stream()
        # (flag mode builds the next batch's blocks on the third stream, beside the later layers' candidate work: one scratch set
        # per layer -- a layer's dense maps live until its block is built)
        sampler.ops.eng.scratch_sets = max(sampler.ops.eng.scratch_sets, len(sampler.nodes_per_layer))
        self.slot, self.primed = 0, False
        self.blocks2 = [None, None]
        self.ev_f, self.ev_b = torch.cuda.Event(), torch.cuda.Event()
        self.g_main, self.g_fx, self.g_s, self.g_b, self.g_blk = [None, None], [None, None], [None, None], [None, None], [None, None]
        self._held = [None, None]
        self._flags_primed, self.use_flags, self.use_third, self.late_block, self.split_output = False, False, False, False, False
        self.late_all = False

    # ---- the three parts ---------------------------------------------------------------------------------------------------
    def _sample(self, slot):
        self._gather_seeds()
        self.blocks2[slot] = self.sampler.enqueue(self.seeds_g, slot=slot)

    def _fwd_x(self, slot):
        blocks = self.blocks2[slot]
        pred = self._forward(blocks, slot)
        self.sampler.exp3(blocks)
        return pred

    FLAG_BLK_DONE, FLAG_F_DONE, FLAG_B_DONE = 11, 12, 13        # slots of the engine's device flags; 0 .. L-1: "layer n's kept list is final"
    FLAG_SEL_DONE = 10                                          # "the next batch's kept lists are final" (late-block mode)

    def _flags_usable(self, tries=4):
        """Device flags order two streams only if the streams really run side by side (HIP multiplexes streams onto a few
        hardware queues; a wait that shares its producer's queue would spin until its time-out): probe with harmless kernels, in
        both directions, and take another stream of PyTorch's pool if the probe fails (as train.PipelinedTrainStep does)."""
        eng = self.sampler.ops.eng
        main = torch.cuda.current_stream()

        def probe(waiter, raiser):
            torch.cuda.synchronize()
            eng.flag_err.zero_(); eng.flags.zero_()
            torch.cuda.synchronize()
            f = eng.flags.data_ptr() + 4 * 14
            _lib.check(_lib.lib.bliss_flag_wait(f, eng.flag_err.data_ptr(), waiter.cuda_stream), "bliss_flag_wait")
            _lib.check(_lib.lib.bliss_flag_raise(f, raiser.cuda_stream), "bliss_flag_raise")
            torch.cuda.synchronize()
            ok = int(eng.flag_err.item()) == 0
            eng.flag_err.zero_(); eng.flags.zero_()
            torch.cuda.synchronize()
            return ok

        for name in ("side", "third"):
            for _ in range(tries):
                st = getattr(self, name)
                # (the third stream's waits must not sit on the backward stream's queue either: a spinning wait there holds B back)
                if probe(st, main) and probe(main, st) and (name == "side" or (probe(st, self.side) and probe(self.side, st))):
                    break
                setattr(self, name, torch.cuda.Stream())
            else:
                return False
        return True

    def _flag(self, which, raise_):
        eng = self.sampler.ops.eng
        st = torch.cuda.current_stream().cuda_stream
        if raise_:
            _lib.check(_lib.lib.bliss_flag_raise(eng.flags.data_ptr() + 4 * which, st), "bliss_flag_raise")
        else:
            _lib.check(_lib.lib.bliss_flag_wait(eng.flags.data_ptr() + 4 * which, eng.flag_err.data_ptr(), st), "bliss_flag_wait")

    def _bwd(self, pred, slot):
        if isinstance(pred, tuple) and len(pred) == 2 and isinstance(pred[0], str):      # (the output layer was left to this stream)
            pred = self._forward_output(self.blocks2[slot], pred, slot)
        out = self._loss_backward_step(self.blocks2[slot], pred, slot)
        self.last = dict(mfgs=self.blocks2[slot], pred=out, slot=slot)

    def prime(self, my_seeds):
        """Sample the first batch (slot 0): the loop trains a batch one call after it was sampled."""
        self.my_seeds.copy_(my_seeds.to(torch.int32), non_blocking=True)
        if self.graph is not None:
            self.g_s[0].replay()
        else:
            self._sample(0)
        self.slot, self.primed = 0, True
        self._flags_primed = False

    def __call__(self, next_seeds):
        """Train the batch sampled by the previous call (or ``prime``) and sample ``next_seeds`` for the next one.  Returns the
        device scalar that holds the trained batch's global mean loss once the backward stream has finished it (``finish``)."""
        if not self.primed:
            raise RuntimeError("prime(seeds) first: the loop trains a batch one call after sampling it")
        s = self.slot
        main = torch.cuda.current_stream()
        self.my_seeds.copy_(next_seeds.to(torch.int32), non_blocking=True)
        if self.graph is not None and self.use_flags:
            # two graphs per step, ordered by DEVICE FLAGS (bliss_flag_wait / raise, bounded spins): the main graph waits for
            # "batch t-1's backward pass and Adam are done" with its first kernel and raises "forward done" behind F; the backward
            # graph waits for that one and raises the other.  Both are launched ahead by the host; a stream-event wait in front of
            # each graph cost 40-90 us per boundary (profiles/r03_u: 1.10 ms/step with three event-ordered graphs)
            if not self._flags_primed:                           # nothing precedes the first step; its blocks were built by prime()
                self._flag(self.FLAG_B_DONE, True)
                if self.use_third or self.late_block:
                    self._flag(self.FLAG_BLK_DONE, True)
                self._flags_primed = True
            self.g_main[s].replay()
            with torch.cuda.stream(self.side):
                self.g_b[s].replay()
            if self.use_third:
                with torch.cuda.stream(self.third):
                    self.g_blk[1 - s].replay()
        elif self.graph is not None:                             # three graphs per step, ordered by stream events
            main.wait_event(self.ev_b)
            self.g_fx[s].replay()
            self.ev_f.record(main)
            self.g_s[1 - s].replay()
            self.side.wait_event(self.ev_f)
            with torch.cuda.stream(self.side):
                self.g_b[s].replay()
                self.ev_b.record(self.side)
        else:
            main.wait_event(self.ev_b)                           # batch t-1: parameters updated, its block slot free again
            pred = self._fwd_x(s)
            self.ev_f.record(main)
            self._sample(1 - s)
            self.side.wait_event(self.ev_f)
            with torch.cuda.stream(self.side):
                self._bwd(pred, s)
                self.ev_b.record(self.side)
        self.trained_slot, self.slot = s, 1 - s
        return self.loss_dev

    def finish(self):
        """Wait for both streams; the trained batch's loss, the sizes of the batch just sampled, the error check."""
        self.side.synchronize()
        self.third.synchronize()
        torch.cuda.current_stream().synchronize()
        sizes = self.sampler.finish(self.slot)
        self.sampler.check_errors()
        self._check_loss_errors()
        if self.graph is not None and self.use_flags and int(self.sampler.ops.eng.flag_err.item()):
            raise RuntimeError("a cross-stream flag never arrived (bliss_flag_wait timed out): the pipelined results are invalid")
        return float(self.loss_dev.item()), sizes

    def calibrate(self, loader, steps=4, k_margin=1.4, b_margin=2.0):
        raise NotImplementedError("capacities for the pipelined loop: shard_static.measure_caps + DenseShardedSampler(fixed_caps=...)")

    def capture(self, loader, warmup=2):
        """Eager pipelined warm-up, then the six graphs (per slot: F + X, S, B; F + X and B share a memory pool: the backward pass
        is recorded against the forward's autograd graph)."""
        if self.g.world > 1 and dist.get_backend(self.group) != "nccl":
            raise RuntimeError("graph capture needs the collectives on the device (backend nccl)")
        import gc
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            if not self.primed:
                self.prime(next(loader))
            for _ in range(warmup):
                self(next(loader))
            self.side.synchronize()
            torch.cuda.current_stream().synchronize()
        torch.cuda.current_stream().wait_stream(warm)
        self.sampler.check_errors()
        self.last = {}
        gc.collect()
        torch.cuda.synchronize()
        pool, pool_b, pool_k = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()                                # (one capture stream for F and B: autograd replays a node on its forward's stream)
        g_main, g_s, g_b, g_blk = [None, None], [None, None], [None, None], [None, None]
        for s in (0, 1):                                         # (S alone: what prime() replays)
            g_s[s] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_s[s], stream=cap, **_cap_kw()):
                self._sample(s)
        import os
        self.use_flags = os.environ.get("BLISS_SHARD_FLAGS", "1") != "0" and self._flags_usable()
        if self.use_flags and self.g.world > 1:
            # the waits of this loop sit in front of work that contains collectives: a peer's hiccup of a second must not read as
            # "the flag never came" (the bound is a kernel argument: it is captured with the graphs below)
            _lib.check(_lib.lib.bliss_flag_set_spin_bound(1 << 27), "bliss_flag_set_spin_bound")
        # BLISS_SHARD_THIRD=1: the next batch's blocks as a graph of their own on a third stream, layer by layer behind "layer n's kept
        # list is final" flags (the single-GPU loop's arrangement).  Correct (same bits) but 2.6 ms/step instead of 1.02 here: the
        # spinning waits of that graph sit on a hardware queue the backward pass needs -- measured, off by default
        self.use_third = self.use_flags and os.environ.get("BLISS_SHARD_THIRD", "0") == "1"
        # BLISS_SHARD_LATE_BLOCK (default on): the input-most block of batch t+1 -- the last and largest of the sampler -- is built on
        # the BACKWARD stream, behind B(t) and a "kept lists final" flag, while the main stream already runs the part of F(t+1) that
        # needs only the kept list (feature rows, halo sum, the input layer's two Linears); F's first aggregation waits for the
        # block (nn._wait_block), as in the single-GPU loop.  Two streams, two graphs per step as before
        # (multi-label runs keep the loss on torch ops: their backward stream is the longer chain already -- Yelp-like 1549 steps/s
        # with all blocks there against 1613 with the input-most block only -- so they default to 1)
        late_mode = os.environ.get("BLISS_SHARD_LATE_BLOCK", "1" if self.multilabel else "2")
        self.late_block = self.use_flags and not self.use_third and late_mode != "0"
        # BLISS_SHARD_LATE_BLOCK=2 (default): ALL blocks of batch t+1 go to the backward stream, block n behind "layer n's kept list is
        # final" (raised by that layer's select launch itself): the backward stream had the room since the loss kernel, and the
        # critical stream keeps only the candidate chain.  =1: the input-most block only
        self.late_all = self.late_block and late_mode in ("2", "3")
        late_from = 1 if late_mode == "3" else 0                 # (=3: the output block stays on the critical stream)
        L_s = len(self.sampler.nodes_per_layer)
        eng = self.sampler.ops.eng
        # BLISS_SHARD_SPLIT_OUTPUT (default on): the forward split of the single-GPU loop (section 6 item 16) -- the EXP3 update reads the
        # blocks' INPUT row norms and nothing the output layer computes, so the critical stream goes from the output layer's input
        # rows straight to X(t) and S(t+1); the output layer itself runs in front of the loss on the backward stream
        self.split_output = self.use_flags and L_s > 1 and os.environ.get("BLISS_SHARD_SPLIT_OUTPUT", "1") != "0"
        g_fx = [None, None]
        for s in (0, 1):
            g_b[s] = torch.cuda.CUDAGraph()
            if self.use_flags:
                g_main[s] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_main[s], pool=pool, stream=cap, **_cap_kw()):
                    self._flag(self.FLAG_B_DONE, False)          # batch t-1: parameters updated, its block slot free again
                    if self.use_third:
                        self._flag(self.FLAG_BLK_DONE, False)    # batch t's blocks are built (third stream, during step t-1)
                    blocks = self.blocks2[s]
                    if self.late_block:                          # (consumed by the first aggregation of F: a wait for BLK_DONE)
                        blocks[0]._ready = (eng.flags.data_ptr() + 4 * self.FLAG_BLK_DONE, eng.flag_err.data_ptr())
                    self._held[s] = self._forward(blocks, s, defer_output=self.split_output)     # F(t)
                    if self.late_block and getattr(blocks[0], "_ready", None) is not None:
                        raise RuntimeError("late-block mode: the forward pass never waited for the input block")
                    self._flag(self.FLAG_F_DONE, True)
                    self.sampler.exp3(blocks)                    # X(t)
                    if self.late_block and self.late_all:        # S(t+1) without its blocks: flag n = "layer n's kept list is final"
                        self._gather_seeds()
                        self.blocks2[1 - s] = self.sampler.enqueue(self.seeds_g, slot=1 - s, defer=tuple(range(late_from, L_s)),
                                                                   layer_done_flags=eng.flags.data_ptr())
                    elif self.late_block:                        # S(t+1) but for its last block
                        self._gather_seeds()
                        self.blocks2[1 - s] = self.sampler.enqueue(self.seeds_g, slot=1 - s, defer=(L_s - 1,),
                                                                   sel_done_flag=eng.flags.data_ptr() + 4 * self.FLAG_SEL_DONE)
                    elif self.use_third:                           # S(t+1) without its blocks; "layer n's kept list is final": flag n
                        self._gather_seeds()
                        self.blocks2[1 - s] = self.sampler.enqueue(self.seeds_g, slot=1 - s, part="select", hook=lambda n: self._flag(n, True))
                    else:
                        self._sample(1 - s)                      # S(t+1)
                if self.use_third:
                    g_blk[1 - s] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_blk[1 - s], pool=pool_k, stream=cap, **_cap_kw()):     # the blocks of batch t+1, third stream
                        self.sampler.enqueue(self.seeds_g, slot=1 - s, part="build", hook=lambda n: self._flag(n, False))
                        self._flag(self.FLAG_BLK_DONE, True)
                # (a pool of its own: B(t) runs BESIDE the S(t+1) part of the main graph, and two graphs that share a pool share the
                # memory of their temporaries -- the sampler's were overwritten by the backward pass's until the pools were split.
                # The forward's saved tensors live in the main graph's pool and stay alive through _held.)
                with torch.cuda.graph(g_b[s], pool=pool_b, stream=cap, **_cap_kw()):
                    self._flag(self.FLAG_F_DONE, False)
                    self._bwd(self._held[s], s)                  # B(t)
                    self._flag(self.FLAG_B_DONE, True)
                    if self.late_block and self.late_all:        # all blocks of batch t+1, each behind its layer's flag
                        self.sampler.enqueue(self.seeds_g, slot=1 - s, part="build", hook=lambda n: self._flag(n, False),
                                             layers=tuple(range(late_from, L_s)), ready_flag=eng.flags.data_ptr() + 4 * self.FLAG_BLK_DONE)
                    elif self.late_block:                        # the input-most block of batch t+1
                        self._flag(self.FLAG_SEL_DONE, False)
                        # (BLK_DONE is raised by bliss_build_block itself, in front of the by-source lists only B(t+1) reads -- this stream)
                        self.sampler.enqueue(self.seeds_g, slot=1 - s, part="build", layers=(L_s - 1,),
                                             ready_flag=eng.flags.data_ptr() + 4 * self.FLAG_BLK_DONE)
            else:
                g_fx[s] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fx[s], pool=pool, stream=cap, **_cap_kw()):
                    self._held[s] = self._fwd_x(s)
                with torch.cuda.graph(g_b[s], pool=pool, stream=cap, **_cap_kw()):
                    self._bwd(self._held[s], s)
        self.g_main, self.g_fx, self.g_s, self.g_b, self.g_blk = g_main, g_fx, g_s, g_b, g_blk
        self.graph = True
        # the captures executed nothing, and the slot the loop trains next now consists of recorded tensors: sample its batch again
        # (my_seeds still holds it) under the SAME step number of the keyed draw -- the loop continues as if nothing had happened
        self.sampler._bufs["step"].sub_(1)
        self.g_s[self.slot].replay()
        torch.cuda.synchronize()

    def close(self):
        import gc
        torch.cuda.synchronize()
        self.g_main, self.g_fx, self.g_s, self.g_b, self.g_blk, self._held = ([None, None] for _ in range(6))
        self.graph, self.last, self.blocks2 = None, {}, [None, None]
        gc.collect()
        torch.cuda.synchronize()
