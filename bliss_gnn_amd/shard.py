"""Destination-range sharding of the sampler, the bandit state and the model's message passing.

The reference is single-device (train_lightning.py:648-657); this is the multi-GPU split its data structures suggest:

  * nodes are cut into contiguous DESTINATION ranges with equal in-edge counts (``partition_by_in_edges``); rank r owns the
    CSC columns of its range, the EXP3 weights of those in-edges (``exp3_weights`` is per in-edge, bandit_sampler.py:342-343:
    reads :127 and updates :248 are owner-local) and the feature rows of its range;
  * one sampling layer = one exchange round:
      1. every rank expands the columns of the seeds it owns (frontier, sum_j w_ij, q_ij, sum_k q_ik: all column-local,
         bandit_sampler.py:123-137, :67) and reduces (q/sum q)^2 by SOURCE over its own edges -- partial sums, :71-73;
      2. (source, partial) pairs travel to the source's owner (all-to-all), which adds them up.  The partials are exact
         Q.44 integers (csrc/common.cuh), so the sum has the same bits for any number of shards;
      3. the histogram of the importances' bit patterns is all-reduced; every rank then runs the same fixed point for the
         Poisson scale c (:391-401) -- one collective instead of one per iteration;
      4. the owner draws its candidates (:403-406, :422-424) with a counter-based uniform keyed by (seed, step, layer, node
         id) -- shards cannot share torch's serial CPU stream -- and the kept (node, P) lists are all-gathered;
      5. every rank builds the block of ITS seeds (:269-339: Hajek weights are per destination) over the global kept list;
  * the model: block inputs are all-gathered (halo features / activations), each rank computes the rows of its own
    destinations, gradients flow back through the gather as a reduce-scatter, parameters are all-reduced;
  * ``exp3``: owner-local updates; the L1 norm of a weight row (:249) is the all-reduced exact row sum.

Parity: the oracle run in the same keyed mode (``oracle.bliss_oracle.sample_blocks_bandit(uniform_fn=keyed_uniform)``) --
same kept sets, same P, same block edges and weights, bit for bit; node ORDER inside a block differs by construction
(seeds first in seed order, then the other kept nodes by owner and node id, instead of first appearance).

The collectives are plain ``torch.distributed`` calls (backend "nccl" = RCCL over xGMI; gloo in the CPU tests).  The
per-edge work is the HIP kernels of the single-GPU sampler behind ``_HipShardOps``; the per-candidate list handling
between two exchanges uses torch tensor ops on the device.  This path launches eagerly (one host sync per exchange, as the
collectives' sizes are data-dependent); the static-shape / HIP-graph treatment of the single-GPU step is future work.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .graph import Block, Graph, NID

HIST_BINS = 32768


# ----------------------------------------------------------------------------------------------- partition
def partition_by_in_edges(indptr, world):
    """Contiguous node ranges with (nearly) equal in-edge counts: ``bounds`` int64 [world+1], rank r owns [bounds[r],
    bounds[r+1]).  Balanced by edges, not nodes: frontier work is per in-edge and hub columns are long."""
    indptr = indptr.detach().cpu()
    V, E = indptr.numel() - 1, int(indptr[-1])
    targets = torch.arange(1, world, dtype=torch.float64) * (E / world)
    cuts = torch.searchsorted(indptr.double(), targets).clamp(max=V)
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64), cuts.to(torch.int64), torch.tensor([V])])
    return torch.cummax(bounds, 0).values


def owner_of(ids, bounds):
    """Rank owning every node id (``bounds`` on the ids' device)."""
    return torch.bucketize(ids.long(), bounds[1:-1].contiguous(), right=True)


class GraphShard(Graph):
    """The CSC columns of one destination range: ``indptr`` still spans all |V| columns (the others are empty), so node ids
    stay global everywhere; ``indices`` / ``eid`` hold only the owned in-edges.  ``ndata_owned[k]`` = rows lo..hi-1."""

    def __init__(self, indptr, indices, eid, bounds, rank, ndata_owned=None):
        super().__init__(indptr, indices, eid)
        self.bounds = bounds.to(indptr.device)
        self.rank, self.world = int(rank), int(bounds.numel() - 1)
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.ndata_owned = dict(ndata_owned or {})

    # ``eid`` holds GLOBAL edge ids (what a block reports as edata[EID]); per-edge tensors of a shard (edata['w'], the EXP3
    # rows) are indexed by LOCAL CSC position, so the edge-id <-> position plumbing of Graph is the identity here
    def by_position(self, edge_tensor):
        return edge_tensor

    def by_edge_id(self, pos_tensor):
        return pos_tensor

    @staticmethod
    def from_global(indptr, indices, eid, bounds, rank, device=None, ndata=None):
        """Cut rank ``rank``'s shard out of a whole CSC (host or device tensors)."""
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        e0, e1 = int(indptr[lo]), int(indptr[hi])
        ip = torch.empty_like(indptr)
        ip[:lo] = 0
        ip[lo:hi + 1] = indptr[lo:hi + 1] - e0
        ip[hi + 1:] = e1 - e0
        dev = indptr.device if device is None else torch.device(device)
        owned = {k: v[lo:hi].to(dev) for k, v in (ndata or {}).items()}
        return GraphShard(ip.to(dev), indices[e0:e1].contiguous().to(dev), None if eid is None else eid[e0:e1].contiguous().to(dev),
                          bounds, rank, owned)


class ShardBlock(Block):
    """The block of one rank: destinations = the seeds this rank owns (``dst_pos`` = their positions in the global source
    list, whose first entries are ALL seeds in seed order); sources index the global kept list."""

    def __init__(self, g, n_src, n_dst, indptr, src, dst, pos, eid, src_nid, dst_pos, dst_nid=None):
        super().__init__(g, n_src, n_dst, indptr, src, dst, pos, eid, src_nid)
        self.dst_pos = dst_pos
        self.dstdata[NID] = src_nid[dst_pos] if dst_nid is None else dst_nid      # (dst_nid: the caller's persistent buffer)


# ----------------------------------------------------------------------------------------------- collectives
def _xdev(t, group=None):
    """Where a collective on ``group`` wants its tensors: RCCL ("nccl") takes device tensors, gloo host tensors (the
    two-ranks-on-one-GPU test and the CPU tests run the same code over gloo)."""
    return t.device if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _all_reduce(t, group=None):
    x = _xdev(t, group)
    if x == t.device:
        dist.all_reduce(t, group=group)
    else:
        tmp = t.to(x)
        dist.all_reduce(tmp, group=group)
        t.copy_(tmp)
    return t


def _all_to_all_by_owner(owner, tensors, world, group=None):
    """Send row i of every tensor to rank owner[i]; returns the received rows (concatenated in rank order)."""
    x = _xdev(owner, group)
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world).to(x)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    ss, rs = send_counts.tolist(), recv_counts.tolist()
    out = []
    for t in tensors:
        src = t[order].contiguous().to(x)
        r = torch.empty((sum(rs),) + tuple(t.shape[1:]), dtype=t.dtype, device=x)
        dist.all_to_all_single(r, src, rs, ss, group=group)
        out.append(r.to(t.device))
    return out


def _all_gather_var(t, world, group=None):
    """All-gather of tensors whose first dimension differs per rank; returns the list per rank."""
    x = _xdev(t, group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=x)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=x)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:s].to(t.device) for o, s in zip(out, sizes)]


class _GatherRows(torch.autograd.Function):
    """Every rank contributes the rows it computed; everyone receives all rows, placed at ``positions`` (the halo exchange
    of block inputs).  Backward: the gradient of a row goes back to the rank that produced it, summed over all consumers
    (all-reduce of the full gradient, own rows sliced: the reduce-scatter of this gather)."""

    @staticmethod
    def forward(ctx, rows, positions, n_total, world, group):
        parts = _all_gather_var(rows, world, group)
        poss = _all_gather_var(positions, world, group)
        full = torch.zeros((n_total,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        for p, r in zip(poss, parts):
            full[p.long()] = r
        ctx.positions, ctx.group = positions, group
        return full

    @staticmethod
    def backward(ctx, gfull):
        g = gfull.float().contiguous()                       # bf16 is not a gloo reduction type; fp32 sums are also the safer choice
        _all_reduce(g, ctx.group)
        return g[ctx.positions.long()].to(gfull.dtype), None, None, None, None


def gather_rows(rows, positions, n_total, world, group=None):
    return _GatherRows.apply(rows, positions, n_total, world, group)


# ----------------------------------------------------------------------------------------------- local ops (HIP)
class _HipShardOps:
    """The per-edge and per-candidate work of one shard on its GPU: the single-GPU sampler's kernels
    (bliss_frontier_prob in BLISS_MODE_PARTIALS, bliss_build_block, bliss_exp3_update) plus csrc/shard.hip."""

    def __init__(self, shard, n_layers, eta, importance_sampling=True, model="sage"):
        from ._engine import LayerEngine
        self.g, self.L, self.eta = shard, n_layers, float(eta)
        self.mode = _lib.MODE_BANDIT | (0 if importance_sampling else _lib.MODE_UNIFORM_NODES)
        self.uniform_nodes = 0 if importance_sampling else 1
        self.eng = LayerEngine(shard)
        if not self.eng.n_bins:
            raise RuntimeError("sharded sampling needs the LDS-binned candidate pipeline (|V| / 1024 node slots in 64 KiB)")
        dev, E = shard.device, shard.num_edges()
        self.w_pos = torch.ones(n_layers, E, dtype=torch.bfloat16, device=dev)            # bandit_sampler.py:342-343 (owned in-edges)
        self.row_sum = torch.zeros(n_layers, 96, dtype=torch.int64, device=dev)
        self.row_sum[:, 2] = E
        self.scratch = torch.zeros(n_layers, 98, dtype=torch.int64, device=dev)
        self.norms = torch.zeros(n_layers, dtype=torch.bfloat16, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cnt = torch.zeros(10, dtype=torch.int32, device=dev)                          # a LayerCounts record for scale / select
        self.sel_scratch = torch.zeros(shard.num_nodes() // 1024 + 4, dtype=torch.int32, device=dev)
        self.edge_w_pos = None
        self._layer = {}

    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def set_caps(self, s_global, fanouts):
        """Capacities per sampling layer from the GLOBAL sizes (the kept list of a layer is global; a rank's seeds are a
        subset of the global seed list)."""
        eng, caps, s = self.eng, [], int(s_global)
        fixed = getattr(self, "fixed_caps", None)          # (shard_static: capacities calibrated from observed sizes, agreed by all ranks)
        for n, f in enumerate(fanouts):
            k = min(eng.V, 2 * (int(f) + s) + 64)
            b = int(min(max(eng.Eg, 1), max(1 << 16, 64 * k)))
            if fixed is not None and len(fixed) == len(fanouts):
                k, b = min(k, int(fixed[n]["K"])), min(b, int(fixed[n]["B"]))
            caps.append(dict(S=s, C=eng.V, K=k, B=b))
            s = k
        if fixed is not None and len(fixed) == len(fanouts):
            if eng.caps is None or len(eng.caps) != len(caps) or any(a[x] != b[x] for a, b in zip(eng.caps, caps) for x in ("S", "K", "B")):
                eng.caps, eng.ws = caps, None
            eng._ensure(int(s_global), fanouts)
            return
        if eng.caps is None or any(a[x] < b[x] for a, b in zip(eng.caps, caps) for x in ("S", "K", "B")) or len(eng.caps) != len(caps):
            eng.caps, eng.ws = caps, None
        eng._ensure(int(s_global), fanouts)

    def frontier_partials(self, n, layer, seeds_local):
        """Sampling layer n over this rank's seeds: (global source ids [m], exact Q.44 partial sums of (q/sum q)^2 [m])."""
        eng, dev = self.eng, self.g.device
        counts = torch.zeros(eng_counts_words(len(eng.caps)), dtype=torch.int32, device=dev)
        c_ws, c_out, lay, cnt_ptr, kept_nid = eng._layer_buffers(n, counts)
        S = int(seeds_local.numel())
        self._layer[n] = dict(c_ws=c_ws, c_out=c_out, lay=lay, counts=counts, kept_nid=kept_nid, seeds=seeds_local, layer=layer)
        if S == 0:
            return torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)
        eta_f, ome_f = float(np.float32(self.eta)), float(np.float32(1.0 - self.eta))
        _lib.check(_lib.lib.bliss_frontier_prob(C.byref(eng.c_graph), C.byref(eng._set(n)["c_maps"]), self.w_pos[layer].data_ptr(),
                                                seeds_local.data_ptr(), S, 0, eng.caps[n]["S"], self.mode | _lib.MODE_PARTIALS,
                                                eta_f, ome_f, eng.Eg, C.byref(c_ws), self._st()), "bliss_frontier_prob")
        b = eng._bin_buffers()
        n_touched = int(b["cursor"][eng.n_bins])                                  # (sync) sources that are not local seeds
        err = int(counts[10 * n + 5])
        if err:
            raise RuntimeError(f"sampler kernel error 0x{err:x}: {_lib.err_string(err)}")
        cs = eng.caps[n]["S"]
        seed_p2 = eng.ws[n].seed_acc.view(torch.int64)[4 * cs: 4 * cs + S]
        ids = torch.cat([seeds_local, (b["tkey"][:n_touched] & 0xFFFFFFFF).to(torch.int32)])
        sums = torch.cat([seed_p2, b["tsum"][:n_touched]])
        return ids, sums

    def importance(self, sums):
        p = torch.empty(sums.numel(), dtype=torch.bfloat16, device=sums.device)
        _lib.check(_lib.lib.bliss_cand_importance(sums.contiguous().data_ptr(), int(sums.numel()), self.uniform_nodes, p.data_ptr(),
                                                  self.err.data_ptr(), self._st()), "bliss_cand_importance")
        return p

    def scale(self, hist, n_cand, fanout, eps=0.9999):
        """The Poisson scale from the GLOBAL histogram / candidate count; kept on the device for keyed_select."""
        h = hist.to(torch.int32).contiguous()
        self.cnt.zero_()
        self.cnt[2] = int(n_cand)
        if self.sel_scratch.numel() < n_cand // 1024 + 4:
            self.sel_scratch = torch.zeros(n_cand // 1024 + 4, dtype=torch.int32, device=h.device)
        _lib.check(_lib.lib.bliss_poisson_scale(h.data_ptr(), self.cnt.data_ptr(), int(fanout), float(eps), self.sel_scratch.data_ptr(),
                                                self._st()), "bliss_poisson_scale")

    def scale_result(self):
        raw = self.cnt.cpu().numpy()
        return float(raw[8:10].view(np.float64)[0]), bool(raw[7]), int(raw[6])

    def keyed_select(self, ids, p, is_seed, seed, step, n):
        P = torch.empty_like(p)
        keep = torch.empty(ids.numel(), dtype=torch.uint8, device=ids.device)
        _lib.check(_lib.lib.bliss_keyed_select(ids.contiguous().data_ptr(), p.data_ptr(), is_seed.to(torch.uint8).contiguous().data_ptr(),
                                               int(ids.numel()), self.cnt.data_ptr(), int(seed), int(step), int(n), P.data_ptr(),
                                               keep.data_ptr(), self._st()), "bliss_keyed_select")
        return P, keep.bool()

    def build_block(self, n, kept_nid_g, node_prob_g, seed_pos):
        """generate_block (bandit_sampler.py:269-339) for this rank's seeds over the global kept list."""
        eng, dev, st = self.eng, self.g.device, self._layer[n]
        seeds, S, K = st["seeds"], int(st["seeds"].numel()), int(kept_nid_g.numel())
        b_indptr, b_src, b_dst, b_pos, b_eid, b_w, b_q, kept_nid, node_prob, cdev, t_indptr, t_edge = st["lay"]
        if S == 0:
            z = torch.zeros(0, dtype=torch.int32, device=dev)
            blk = ShardBlock(self.g, K, 0, torch.zeros(1, dtype=torch.int32, device=dev), z, z, z, z, kept_nid_g, seed_pos)
            blk._edge_weights = blk._q = torch.zeros(0, dtype=torch.bfloat16, device=dev)
            blk._node_prob, blk._counts_dev, blk._layer = node_prob_g, st["counts"][10 * n:10 * n + 10], st["layer"]
            blk._transposed = (torch.zeros(K + 1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
            return blk
        if K > eng.caps[n]["K"]:
            raise RuntimeError("kept-node capacity exceeded in the sharded sampler")
        sset = eng._set(n)
        kept_nid[:K] = kept_nid_g
        node_prob[:K] = node_prob_g
        sset["kept_map"][kept_nid_g.long()] = torch.arange(K, dtype=torch.int32, device=dev)
        eng.ws[n].cand_nid[:S] = seeds                                     # what the cleanup pass resets in local_id
        cdev[2], cdev[3] = S, K                                            # LayerCounts::C (cleanup bound), ::K
        eta_f, ome_f = float(np.float32(self.eta)), float(np.float32(1.0 - self.eta))
        _lib.check(_lib.lib.bliss_build_block(C.byref(eng.c_graph), C.byref(sset["c_maps"]), self.w_pos[st["layer"]].data_ptr(),
                                              seeds.data_ptr(), eng.caps[n]["S"], self.mode, eta_f, ome_f, eng.Eg, C.byref(st["c_ws"]),
                                              C.byref(st["c_out"]), self._st()), "bliss_build_block")
        raw = cdev.cpu().numpy()                                            # (sync)
        B, err = int(raw[4]), int(raw[5])
        if err:
            raise RuntimeError(f"sharded block build error 0x{err:x}: {_lib.err_string(err)}")
        blk = ShardBlock(self.g, K, S, b_indptr[:S + 1], b_src[:B], b_dst[:B], b_pos[:B], b_eid[:B], kept_nid[:K].clone(), seed_pos)
        blk._edge_weights, blk._q, blk._node_prob = b_w[:B], b_q[:B], node_prob[:K].clone()
        blk._counts_dev, blk._layer = cdev, st["layer"]
        if t_indptr is not None:
            blk._transposed = (t_indptr[:K + 1], t_edge[:max(B, 1)])
        return blk

    def exp3_update(self, blk, embed_norm, delta_f):
        """calculate_rewards + the multiplicative update (bandit_sampler.py:160-248) on the owned in-edges of ``blk``."""
        g = self.g
        if blk.num_edges() == 0:
            return
        if self.edge_w_pos is None:
            from .bandit_sampler import normalized_edata
            self.edge_w_pos = g.by_position(normalized_edata(g)).contiguous()
        B = blk.num_edges()
        rewards = torch.empty(B, dtype=torch.bfloat16, device=g.device)
        idx = blk._layer
        _lib.check(_lib.lib.bliss_exp3_update(
            C.byref(self.eng.c_graph), self.edge_w_pos.data_ptr(), self.w_pos[idx].data_ptr(), self.row_sum[idx].data_ptr(),
            blk.indptr.data_ptr(), blk.src.data_ptr(), blk.dst.data_ptr(), blk.pos.data_ptr(), blk._q.data_ptr(),
            blk._node_prob.data_ptr(), embed_norm.contiguous().data_ptr(), 0, blk.dstdata[NID].contiguous().data_ptr(), blk.num_dst_nodes(),
            blk._counts_dev.data_ptr() + 16, B, delta_f, rewards.data_ptr(), 0, 1, self.err.data_ptr(), self._st()), "bliss_exp3_update")
        blk.edata["rewards"] = rewards

    def exp3_update_all(self, blks, delta_f):
        """exp3_update for all blocks of a step in ONE launch (bliss_exp3_update_blocks; the layers' rows are disjoint)."""
        g = self.g
        if self.edge_w_pos is None:
            from .bandit_sampler import normalized_edata
            self.edge_w_pos = g.by_position(normalized_edata(g)).contiguous()
        recs = (_lib.Exp3Block * len(blks))()
        keep = []
        for i, blk in enumerate(blks):
            B, idx = blk.num_edges(), blk._layer
            rewards = torch.empty(max(B, 1), dtype=torch.bfloat16, device=g.device)
            en, nid = blk.srcdata["embed_norm"].contiguous(), blk.dstdata[NID].contiguous()
            keep.append((en, nid))
            recs[i] = _lib.Exp3Block(self.w_pos[idx].data_ptr(), self.row_sum[idx].data_ptr(), self.scratch[idx].data_ptr(),
                                     self.norms[idx:].data_ptr(), blk.indptr.data_ptr(), blk.src.data_ptr(), blk.dst.data_ptr(),
                                     blk.pos.data_ptr(), blk._q.data_ptr(), blk._node_prob.data_ptr(), en.data_ptr(), 0, nid.data_ptr(),
                                     blk._counts_dev.data_ptr() + 16, rewards.data_ptr(), B, 0)
            blk.edata["rewards"] = rewards
        _lib.check(_lib.lib.bliss_exp3_update_blocks(C.byref(self.eng.c_graph), self.edge_w_pos.data_ptr(), recs, len(blks), delta_f,
                                                     self.err.data_ptr(), self._st()), "bliss_exp3_update_blocks")

    def normalize(self, idx, group=None):
        """F.normalize(row, p=1) (:249) over the row that is spread over all shards: the norm is the all-reduced exact sum."""
        limbs = _all_reduce(self.row_sum[idx].clone(), group)
        _lib.check(_lib.lib.bliss_exp3_normalize_global(self.w_pos[idx].data_ptr(), self.g.num_edges(), self.row_sum[idx].data_ptr(),
                                                        limbs.data_ptr(), self.scratch[idx].data_ptr(), self.norms[idx:].data_ptr(),
                                                        self._st()), "bliss_exp3_normalize_global")

    def check_errors(self):
        bits = int(self.err.item()) | int((self.scratch[:, 0] >> 20).max().item())
        if bits:
            raise RuntimeError(f"sharded sampler kernel error 0x{bits:x}: {_lib.err_string(bits)}")


def eng_counts_words(n_layers):
    return 10 * n_layers


# ----------------------------------------------------------------------------------------------- the sampler
class ShardedPoissonBanditSampler:
    """PoissonBanditLadiesSampler (bandit_sampler.py:369-425) over destination-range shards.  ``sample_blocks(seeds)``
    takes the GLOBAL seed list (identical on every rank) and returns this rank's blocks; ``ops`` does the local work
    (default: the HIP kernels; the CPU tests plug in the oracle's arithmetic to exercise the exchange logic under gloo)."""

    def __init__(self, shard, nodes_per_layer, eta=0.4, importance_sampling=True, seed=0, model="sage", group=None, ops=None):
        self.g, self.nodes_per_layer, self.eta = shard, list(nodes_per_layer), eta
        self.seed, self.group, self.step = int(seed), group, 0
        self.world, self.rank = shard.world, shard.rank
        self.delta = 0.01                                                   # bandit_sampler.py:233
        self._delta_f = float(torch.tensor(self.delta, dtype=torch.float32))
        self.ops = ops if ops is not None else _HipShardOps(shard, len(self.nodes_per_layer), eta, importance_sampling, model)
        self.trace = []

    def sample_blocks(self, seeds_g, step=None):
        """bandit_sampler.py:341-367 for the global seed list ``seeds_g`` (int32, same on every rank).  Returns
        (input_nodes, output_nodes, blocks): this rank's blocks (input-most first) over the GLOBAL kept lists."""
        g, ops, world, grp = self.g, self.ops, self.world, self.group
        dev = seeds_g.device
        step = self.step if step is None else int(step)
        self.step = step + 1
        seeds_g = seeds_g.to(torch.int32).contiguous()
        L = len(self.nodes_per_layer)
        order = list(reversed(range(L)))                                     # :350
        fan = [self.nodes_per_layer[b] for b in order]
        if hasattr(ops, "set_caps"):
            ops.set_caps(int(seeds_g.numel()), fan)
        bounds = g.bounds
        blocks, self.trace = [], []
        for n, layer in enumerate(order):
            S_g = int(seeds_g.numel())
            mine = (seeds_g >= g.lo) & (seeds_g < g.hi)
            seed_pos = torch.nonzero(mine).flatten()
            seeds_l = seeds_g[seed_pos].contiguous()
            # 1. column-local work + partial by-source sums over my edges
            ids, sums = ops.frontier_partials(n, layer, seeds_l)
            # 2. partials -> source owners (exact integers: the sum is schedule- and shard-independent)
            rid, rsum = _all_to_all_by_owner(owner_of(ids, bounds), [ids, sums], world, grp)
            n_own = g.hi - g.lo
            acc = torch.zeros(n_own, dtype=torch.int64, device=dev)
            acc.index_add_(0, rid.long() - g.lo, rsum)
            touched = torch.zeros(n_own, dtype=torch.bool, device=dev)
            touched[rid.long() - g.lo] = True
            seed_mask = torch.zeros(n_own, dtype=torch.bool, device=dev)
            seed_mask[seeds_l.long() - g.lo] = True                         # seeds are candidates whether or not they are sources
            cand_off = torch.nonzero(touched | seed_mask).flatten()
            cand = (cand_off + g.lo).to(torch.int32)
            is_seed = seed_mask[cand_off]
            p = ops.importance(acc[cand_off])                               # :75 sqrt of the summed squares
            # 3. global histogram of p's bit patterns + candidate count -> the same c on every rank
            hist = torch.zeros(HIST_BINS + 1, dtype=torch.int64, device=dev)
            hist[:HIST_BINS] = torch.bincount(p.view(torch.int16).long() & 0xFFFF, minlength=HIST_BINS)[:HIST_BINS]
            hist[HIST_BINS] = cand.numel()
            _all_reduce(hist, grp)
            C_g = int(hist[HIST_BINS])
            ops.scale(hist[:HIST_BINS], C_g, fan[n])
            # 4. keyed draw at the owner; kept non-seed candidates to everybody
            P, keep = ops.keyed_select(cand, p, is_seed, self.seed, step, n)
            new = keep & ~is_seed
            # (one message per rank: node id | P's bits as a second int32 -- bf16 / int16 are not gloo types)
            got = _all_gather_var(torch.stack([cand[new], P[new].view(torch.int16).to(torch.int32)], 1), world, grp)
            kept_g = torch.cat([seeds_g] + [x[:, 0] for x in got]).contiguous()
            prob_g = torch.cat([torch.full((S_g,), 0x3F80, dtype=torch.int16, device=dev)] +
                               [x[:, 1].to(torch.int16) for x in got]).view(torch.bfloat16)
            # 5. my block over the global kept list
            blk = ops.build_block(n, kept_g, prob_g, seed_pos)
            blk.edata["edge_weights"] = blk._edge_weights                   # :324
            blk.edata["q_ij"] = blk._q                                      # :326
            blk.srcdata["node_prob"] = blk._node_prob                       # :328
            self.trace.append(dict(cand=cand, p=p, P=P, C=C_g, scale=ops.scale_result() if hasattr(ops, "scale_result") else None))
            blocks.insert(0, blk)                                           # :366
            seeds_g = kept_g                                                # :364
        return blocks[0].srcdata[NID], blocks[-1].dstdata[NID], blocks

    def exp3(self, mfgs, g=None):
        """bandit_sampler.py:251-267 on the owned in-edges of every block, then the global L1 renormalisation."""
        for idx, mfg in enumerate(mfgs):
            self.ops.exp3_update(mfg, mfg.srcdata["embed_norm"], self._delta_f)
            self.ops.normalize(idx, self.group)

    def check_errors(self):
        if hasattr(self.ops, "check_errors"):
            self.ops.check_errors()


# ----------------------------------------------------------------------------------------------- the model step
def sharded_sage_forward(model, blocks, x_owned_rows_fn, world, group=None):
    """SAGE.forward (model.py:312-333) over shard blocks: the input rows of every block are gathered from the ranks that
    hold them (features: the owners of the nodes; activations: the ranks that computed them), each rank computes the
    rows of its own destinations.  ``x_owned_rows_fn(node_ids) -> rows`` serves this rank's feature rows."""
    from .nn import embed_norm
    h_rows, pos = None, None
    for l, (layer, blk) in enumerate(zip(model.layers, blocks)):
        K = blk.num_src_nodes()
        if l == 0:
            nid = blk.srcdata[NID]
            mine = torch.nonzero((nid >= blk.g.lo) & (nid < blk.g.hi)).flatten()
            h_rows, pos = x_owned_rows_fn(nid[mine]), mine                  # train_lightning.py:138, owner side
        h_src = gather_rows(h_rows, pos, K, world, group)                   # halo exchange
        blk.srcdata["embed_norm"] = embed_norm(h_src)                       # model.py:318-320
        h = layer(blk, (h_src, h_src[blk.dst_pos]), edge_weight=blk.edata["edge_weights"])
        if l < len(model.layers) - 1:
            h = model.dropout(model.activation(h))                          # :330-332
        h_rows, pos = h, blk.dst_pos                                        # my rows of the next block's input
    return h


def allreduce_gradients_sum(model, scale, group=None):
    """Sum the gradients over ranks (each rank's loss covers its own output seeds) and scale by 1 / global batch."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = _all_reduce(torch.cat([g.reshape(-1).float() for g in grads]), group)
    flat.mul_(scale)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class ShardedTrainStep:
    """One optimiser step of ModelLightning (train_lightning.py:100-168, 205-216, 463-471) over destination-range shards:
    the global batch is the concatenation of every rank's own batch (weak scaling; each rank draws seeds it owns), the
    sampler and the model exchange as described at the top of this module, the loss is the mean over the GLOBAL batch."""

    def __init__(self, shard, sampler, model, lr=0.002, multilabel=False, group=None):
        import torch.nn as nn
        self.g, self.sampler, self.model, self.group = shard, sampler, model, group
        self.loss_fn = nn.BCEWithLogitsLoss(reduction="sum") if multilabel else nn.CrossEntropyLoss(reduction="sum")   # :77-79
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)                                                       # :206
        self.last = {}

    def _owned_rows(self, key):
        feats, lo = self.g.ndata_owned[key], self.g.lo
        return lambda nid: feats[(nid.long() - lo)]

    def __call__(self, my_seeds):
        g, world = self.g, self.g.world
        parts = _all_gather_var(my_seeds.to(torch.int32), world, self.group)
        seeds_g = torch.cat(parts).contiguous()                              # the global batch, rank by rank
        inp, outp, blocks = self.sampler.sample_blocks(seeds_g)
        if any(b.num_dst_nodes() == 0 for b in blocks):
            raise RuntimeError("a rank owns no destination of some block: the sharded model step needs every rank to take part")
        pred = sharded_sage_forward(self.model, blocks, self._owned_rows("features"), world, self.group)             # :138-141
        y = self._owned_rows("labels")(blocks[-1].dstdata[NID])                                                      # :139
        n_global = int(seeds_g.numel())
        scale = 1.0 / (n_global * (pred.shape[1] if isinstance(self.loss_fn, torch.nn.BCEWithLogitsLoss) else 1))
        loss_sum = self.loss_fn(pred.float(), y)
        self.opt.zero_grad(set_to_none=True)
        loss_sum.backward()
        allreduce_gradients_sum(self.model, scale, self.group)
        self.opt.step()
        self.sampler.exp3(blocks)                                                                                     # :469-471
        tot = _all_reduce(loss_sum.detach().float().reshape(1).clone(), self.group)
        self.last = dict(mfgs=blocks, pred=pred, loss=float(tot) * scale)
        return self.last["loss"]
