"""EXP3-bandit LADIES samplers on MI355X -- same names, constructor arguments and call sites as the
reference's ``bandit_sampler.py`` so that ``train_lightning.py:358-370, 469-471`` work unchanged:

    sampler = PoissonBanditLadiesSampler(fanouts, importance_sampling=..., node_embedding='features',
                                         num_steps=..., eta=..., model=...)
    input_nodes, output_nodes, blocks = sampler.sample_blocks(g, seed_nodes)
    ...forward / backward / optimiser step...
    sampler.exp3(blocks, g)

The bodies are hand-written gfx950 kernels behind the C ABI of include/bliss_gnn.h.
"""
import ctypes as C

import torch

from . import _lib
from ._engine import LayerEngine, _stream
from .graph import EID, NID, Graph, as_graph


def find_indices_in(a, b):
    """bandit_sampler.py:5-14 -- kept for API compatibility (the kernels never need it: seeds are
    local ids 0..S-1 by construction)."""
    b_sorted, indices = torch.sort(b)
    sorted_indices = torch.searchsorted(b_sorted, a)
    sorted_indices[sorted_indices >= indices.shape[0]] = 0
    return indices[sorted_indices]


def union(*arrays):
    """bandit_sampler.py:16-18."""
    return torch.unique(torch.cat(arrays))


def normalized_edata(g: Graph, weight=None):
    """bandit_sampler.py:20-27 with weight=None: ``w_e = 1 / indeg(dst(e))`` in bf16, by edge id."""
    if weight is not None:
        raise NotImplementedError("only the weight=None form is used by the reference (train_lightning.py:359,362)")
    w_pos = torch.empty(g.num_edges(), dtype=torch.bfloat16, device=g.device)
    cg = _lib.Graph(g.indptr.data_ptr(), g.indices.data_ptr(), 0, g.num_nodes(), g.num_edges())
    _lib.check(_lib.lib.bliss_normalized_edata(C.byref(cg), w_pos.data_ptr(), _stream()), "bliss_normalized_edata")
    return g.by_edge_id(w_pos)


class BlockSampler:
    """The part of ``dgl.dataloading.BlockSampler`` the DataLoader relies on."""

    def __init__(self, *args, **kwargs):
        pass

    def sample(self, g, seed_nodes, exclude_eids=None):
        return self.sample_blocks(g, seed_nodes, exclude_eids=exclude_eids)


class BanditLadiesSampler(BlockSampler):
    """bandit_sampler.py:29-367.  ``select_neighbors`` (:84-99) is ``torch.multinomial`` on the device-computed
    importances, drawn on the host (one sync per layer, like the reference)."""

    _poisson = False

    def __init__(self, nodes_per_layer, importance_sampling=True, weight="w", out_weight="edge_weights",
                 node_embedding="nfeat", node_prob="node_prob", replace=False, eta=0.4, num_steps=5000,
                 model="sage"):
        super().__init__()
        self.nodes_per_layer = nodes_per_layer
        self.importance_sampling = importance_sampling
        self.edge_weight = weight
        self.output_weight = out_weight
        self.node_prob = node_prob
        self.node_embedding = node_embedding
        self.replace = replace
        self.eta = eta
        self.T = num_steps
        self.model = model
        self.eps = 0.9999
        self.delta = 0.01                      # bandit_sampler.py:233
        self._delta_f = float(torch.tensor(self.delta, dtype=torch.float32))
        self._w_pos = None                     # exp3 weights [L, |E|] bf16 in CSC-position order
        self._row_sum = None                   # exact row sums, int64 [L, 3 * 32 replicas]
        self._engine = None

    def _mode(self):
        return _lib.MODE_BANDIT | (0 if self.importance_sampling else _lib.MODE_UNIFORM_NODES)

    # -- state ------------------------------------------------------------------------------
    def _graph(self, g):
        """The bliss Graph behind ``g`` (a DGLGraph-like object is converted once and cached, graph.as_graph)."""
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        return as_graph(g, self._graphs)

    def _bind(self, g):
        g = self._graph(g)
        if self._engine is None or self._engine.g is not g:
            self._engine = LayerEngine(g)
        return self._engine

    def _ensure_weights(self, g):
        if self._w_pos is None:                                         # bandit_sampler.py:342-343
            L, E = len(self.nodes_per_layer), g.num_edges()
            self._w_pos = torch.ones(L, E, dtype=torch.bfloat16, device=g.device)
            rs = torch.zeros(L, 96, dtype=torch.int64, device=g.device)      # 32 replicas of three limbs, summed by the readers
            rs[:, 2] = E                                                # sum of E ones = E * 2^64
            self._row_sum = rs
            self._scratch = torch.zeros(L, 98, dtype=torch.int64, device=g.device)
            self._err = torch.zeros(1, dtype=torch.int32, device=g.device)
            self._norms = torch.zeros(L, dtype=torch.bfloat16, device=g.device)
            self._pend = torch.zeros(L, 4, dtype=torch.int32, device=g.device)     # bliss_norm_state_t per row

    # -- F.normalize off the critical path ----------------------------------------------------
    # bandit_sampler.py:249 renormalises every row after every update.  Here the pass over a row (4 bytes of HBM traffic per
    # edge of the GRAPH) only runs when the row's bf16 norm is not exactly 1.0 -- which still happens in stretches of hundreds
    # of steps (the bf16 quotients over- and undershoot: norm 1.0078, 0.9961, 1.0078 ... until the drift of the sum ends the
    # cycle), and the next batch's sampler waits for exp3().  With ``defer_normalize`` set (train.PipelinedTrainStep does)
    # exp3() only decides which rows need the pass; ``normalize_pending`` runs it OUT OF PLACE into a second buffer per row
    # (beside the next sampler, which reads the old buffer and divides on the fly until the pass has switched the row over).
    # Same bits as the immediate pass.  Everything outside this protocol sees ``_w_pos`` only: ``_settle`` brings the rows home.
    defer_normalize = False
    _pend_maybe = False            # host-side hints (conservative): a pass may be pending / a row may live in _w_alt
    _alt_maybe = False
    _w_alt = None

    def enable_deferred_normalize(self, on=True):
        if not on:
            self._settle()
            self.defer_normalize = False
            return
        if self._w_pos is None:
            raise RuntimeError("enable_deferred_normalize: the sampler has no weights yet (sample once first)")
        if self._w_alt is None or self._w_alt.shape != self._w_pos.shape:
            self._w_alt = torch.empty_like(self._w_pos)
            d = self._w_alt.data_ptr() - self._w_pos.data_ptr()
            if d % 16:
                raise RuntimeError("allocator returned a buffer off the 16-byte phase of the rows")
            st = torch.zeros(self._w_pos.shape[0], 2, dtype=torch.int64)
            st[:, 1] = d // 2                                          # bliss_norm_state_t::alt, in elements
            self._pend.copy_(st.view(torch.int32).to(self._pend.device))
        self.defer_normalize = True

    def _norm_rows(self):
        L = self._w_pos.shape[0]
        rows = (_lib.Exp3Block * L)()
        for idx in range(L):
            rows[idx].w_pos, rows[idx].row_sum = self._w_pos[idx].data_ptr(), self._row_sum[idx].data_ptr()
            rows[idx].scratch, rows[idx].norm_pend = self._scratch[idx].data_ptr(), self._pend[idx:].data_ptr()
        return rows, L

    def normalize_pending(self, g=None):
        """Run the pass over the rows a deferred exp3() left pending (two launches; nothing happens on the device when there
        are none).  Must not overlap the next update."""
        if self._w_pos is None or self._w_alt is None:
            return
        rows, L = self._norm_rows()
        _lib.check(_lib.lib.bliss_exp3_normalize_pending(rows, L, self._w_pos.shape[1], _stream()), "bliss_exp3_normalize_pending")
        self._pend_maybe, self._alt_maybe = False, True

    def _settle(self):
        """Leave the deferred protocol's state behind: no pass pending, every row in ``_w_pos`` (one host read of the state
        words when a row may have moved; off the hot path)."""
        if self._pend_maybe:
            self.normalize_pending()
        if self._alt_maybe:
            st = self._pend[:, 0].tolist()
            for idx, v in enumerate(st):
                if v & 0x20000:
                    self._w_pos[idx].copy_(self._w_alt[idx])
                    self._pend[idx, 0] = v & ~0x20000
            self._alt_maybe = False

    @property
    def exp3_weights(self):
        """[L, |E|] bf16 indexed by EDGE ID like the reference's attribute (bandit_sampler.py:43)."""
        if self._w_pos is None:
            return None
        self._settle()
        return self._engine.g.by_edge_id(self._w_pos)

    @exp3_weights.setter
    def exp3_weights(self, value):
        if value is None:
            self._w_pos = None
            return
        g = self._engine.g
        self._w_pos = g.by_position(value.to(torch.bfloat16)).contiguous().clone()
        L = self._w_pos.shape[0]
        self._row_sum = torch.zeros(L, 96, dtype=torch.int64, device=g.device)
        self._scratch = torch.zeros(L, 98, dtype=torch.int64, device=g.device)
        self._err = torch.zeros(1, dtype=torch.int32, device=g.device)
        self._norms = torch.zeros(L, dtype=torch.bfloat16, device=g.device)
        self._pend = torch.zeros(L, 4, dtype=torch.int32, device=g.device)
        self._pend_maybe = self._alt_maybe = False
        self._w_alt = None
        self.defer_normalize = False
        for l in range(L):
            _lib.check(_lib.lib.bliss_row_sum(self._w_pos[l].data_ptr(), g.num_edges(), self._row_sum[l].data_ptr(),
                                              _stream()), "bliss_row_sum")

    # -- sampling ---------------------------------------------------------------------------
    def sample_blocks(self, g, seed_nodes, exclude_eids=None, uniforms=None):
        """bandit_sampler.py:341-367.  ``uniforms``: optional list (sampling order, last layer first)
        of fp32 vectors used instead of the global CPU generator."""
        g = self._graph(g)
        eng = self._bind(g)
        self._ensure_weights(g)
        self._settle()
        output_nodes = seed_nodes
        order = list(reversed(range(len(self.nodes_per_layer))))          # :350
        rows, fan = [self._w_pos[b] for b in order], [self.nodes_per_layer[b] for b in order]
        if self._poisson:
            blks = eng.sample_blocks(rows, seed_nodes, fan, self._mode(), self.eta, self.eps, uniforms)
        else:                                                             # select_neighbors :84-99 (torch.multinomial)
            blks = eng.sample_blocks_multinomial(rows, seed_nodes, fan, self._mode(), self.eta, self.replace)
        blocks = []
        for blk in blks:
            blk.edata[self.output_weight] = blk._edge_weights           # :324
            blk.edata["q_ij"] = blk._q                                  # :326
            blk.srcdata[self.node_prob] = blk._node_prob                # :328
            blocks.insert(0, blk)                                       # :366
        return blocks[0].srcdata[NID], output_nodes, blocks             # :364,:367

    def sample_blocks_static(self, g, seed_nodes, slot=0, chain_rng=False, external_rng=False, part=None, last_block=True, ready_flag=0):
        """sample_blocks with capacity-padded (static-shape) blocks and no host round trip: everything is only
        ENQUEUED, so the whole train step can be recorded into a HIP graph.  Call ``engine.stage_rng_from_torch()``
        before and ``finish_static()`` after the stream has been synchronised.  Padded rows / edges are inert:
        ids past the true K point at node 0 and no edge references them; edges past the true B are ignored by
        every kernel (the true counts live on the device)."""
        g = self._graph(g)
        eng = self._bind(g)
        self._ensure_weights(g)
        order = list(reversed(range(len(self.nodes_per_layer))))
        pend = [self._pend[b:].data_ptr() for b in order] if self.defer_normalize else None
        if pend is None:
            self._settle()
        blks = eng.enqueue_static([self._w_pos[b] for b in order], seed_nodes, [self.nodes_per_layer[b] for b in order],
                                  self._mode(), self.eta, self.eps, slot=slot, chain_rng=chain_rng, external_rng=external_rng, part=part,
                                  w_pend=pend, last_block=last_block, ready_flag=ready_flag)
        blocks = []
        for blk in blks:
            blk.edata[self.output_weight] = blk._edge_weights
            blk.edata["q_ij"] = blk._q
            blk.srcdata[self.node_prob] = blk._node_prob
            blocks.insert(0, blk)
        return blocks[0].srcdata[NID], seed_nodes, blocks

    def finish_static(self, slot=0, commit=True):
        return self._engine.finish(slot, commit)

    # -- bandit update ----------------------------------------------------------------------
    def exp3(self, mfgs, g, apply=True, factors=None, bounds=None, done_flag=None):
        """bandit_sampler.py:251-267: rewards + weight update + L1 renormalisation, per block.

        ``apply=False`` (multi-GPU replicas, bliss_gnn_amd/dist.py) only computes the rewards and, into
        ``factors[idx]`` (bf16 [B]), the multiplicative updates; ``apply_updates`` then applies every
        rank's updates in rank order.  ``bounds[idx]``: process at most that many edges of block idx (the length of
        ``factors[idx]``); a longer block is flagged (error bit 8)."""
        g = self._graph(g)
        self._bind(g)
        st = _stream()
        edge_w_pos = g.edata_by_position(self.edge_weight)
        cg = self._engine.c_graph
        fused = apply and factors is None and len(mfgs) <= 8         # all blocks in two launches (bliss_exp3_step)
        defer = fused and self.defer_normalize
        if defer:
            if self._pend_maybe:                                     # (a caller that overlaps the pass has launched it already)
                self.normalize_pending()
        else:
            self._settle()
        keep = []                                                    # (tensors the launch reads must outlive the loop)
        recs = (_lib.Exp3Block * len(mfgs))() if fused else None
        for idx, mfg in enumerate(mfgs):
            B = mfg.num_edges()
            alpha = None
            if self.model == "gat":                                    # calculate_alpha, bandit_sampler.py:146-154
                a_ij = mfg.edata["a_ij"].detach()
                a_ij = a_ij.bfloat16().contiguous() if a_ij.dtype != torch.bfloat16 else a_ij.contiguous()
                alpha = torch.empty(B, dtype=torch.bfloat16, device=g.device)
                _lib.check(_lib.lib.bliss_gat_alpha(mfg.indptr.data_ptr(), mfg.num_dst_nodes(), mfg.edata["q_ij"].data_ptr(),
                                                    a_ij.data_ptr(), alpha.data_ptr(), self._err.data_ptr(), st), "bliss_gat_alpha")
            n_edges_ptr = mfg._counts_dev.data_ptr() + 16             # LayerCounts::B, already on the device
            rewards = torch.empty(B, dtype=torch.bfloat16, device=g.device)
            en = mfg.srcdata["embed_norm"]
            if en.dtype != torch.bfloat16:
                en = en.bfloat16()
            en = en.contiguous()
            if fused:
                keep.append((alpha, en))
                recs[idx] = _lib.Exp3Block(self._w_pos[idx].data_ptr(), self._row_sum[idx].data_ptr(), self._scratch[idx].data_ptr(),
                                           self._norms[idx:].data_ptr(), mfg.indptr.data_ptr(), mfg.src.data_ptr(), mfg.dst.data_ptr(),
                                           mfg.pos.data_ptr(), mfg.edata["q_ij"].data_ptr(), mfg.srcdata[self.node_prob].data_ptr(),
                                           en.data_ptr(), 0 if alpha is None else alpha.data_ptr(), mfg.dstdata[NID].data_ptr(),
                                           n_edges_ptr, rewards.data_ptr(), B, self._pend[idx:].data_ptr())
                mfg.edata["rewards"] = rewards                          # :193
                continue
            _lib.check(_lib.lib.bliss_exp3_update(
                C.byref(cg), edge_w_pos.data_ptr(), self._w_pos[idx].data_ptr(), self._row_sum[idx].data_ptr(),
                mfg.indptr.data_ptr(), mfg.src.data_ptr(), mfg.dst.data_ptr(), mfg.pos.data_ptr(),
                mfg.edata["q_ij"].data_ptr(), mfg.srcdata[self.node_prob].data_ptr(), en.contiguous().data_ptr(),
                0 if alpha is None else alpha.data_ptr(), mfg.dstdata[NID].data_ptr(), mfg.num_dst_nodes(),
                n_edges_ptr, B if bounds is None else min(B, int(bounds[idx])), self._delta_f, rewards.data_ptr(),
                0 if factors is None else factors[idx].data_ptr(), int(apply), self._err.data_ptr(), st), "bliss_exp3_update")
            mfg.edata["rewards"] = rewards                              # :193
            if not apply:
                continue
            _lib.check(_lib.lib.bliss_exp3_normalize(self._w_pos[idx].data_ptr(), g.num_edges(),
                                                     self._row_sum[idx].data_ptr(), self._scratch[idx].data_ptr(),
                                                     self._norms[idx:].data_ptr(), st), "bliss_exp3_normalize")
        if fused and len(mfgs):
            if defer:
                _lib.check(_lib.lib.bliss_exp3_step_deferred(C.byref(cg), edge_w_pos.data_ptr(), recs, len(mfgs), self._delta_f,
                                                             int(done_flag or 0), self._err.data_ptr(), st), "bliss_exp3_step_deferred")
                self._pend_maybe = True
            else:
                _lib.check(_lib.lib.bliss_exp3_step(C.byref(cg), edge_w_pos.data_ptr(), recs, len(mfgs), self._delta_f,
                                                    self._err.data_ptr(), st), "bliss_exp3_step")

    def apply_updates(self, idx, pos, factor, g, n_dev=None):
        """w[pos] *= factor on layer ``idx`` (positions unique within one call), bandit_sampler.py:248.  ``n_dev``: optional
        int32 device tensor holding how many leading entries are valid (capacity-padded lists, graph capture)."""
        bound = int(pos.numel())
        if bound == 0:
            return
        self._settle()
        if n_dev is None:
            n_dev = torch.tensor([bound], dtype=torch.int32, device=pos.device)
        _lib.check(_lib.lib.bliss_exp3_apply(self._w_pos[idx].data_ptr(), self._row_sum[idx].data_ptr(), pos.data_ptr(),
                                             factor.data_ptr(), n_dev.data_ptr(), bound, self._err.data_ptr(), _stream()),
                   "bliss_exp3_apply")

    def apply_updates_ranks(self, block_ids, gathered, rank_stride, n_ranks, pos_off, factor_off_bf16, count_off, bounds):
        """apply_updates for the packed (position, factor, count) lists of ``n_ranks`` ranks and several blocks in ONE launch,
        rank after rank (bliss_exp3_apply_ranks; ``gathered`` int32 = the all-gathered buffers, offsets per block inside one
        rank's buffer).  Same bits as the corresponding apply_updates calls in rank order."""
        self._settle()
        m = _lib.Exp3RankLists()
        for j, idx in enumerate(block_ids):
            m.w_pos[j], m.row_sum[j] = self._w_pos[idx].data_ptr(), self._row_sum[idx].data_ptr()
            m.pos_off_words[j], m.factor_off_bf16[j], m.count_off_words[j], m.bound[j] = pos_off[j], factor_off_bf16[j], count_off[j], bounds[j]
        m.n_blocks, m.n_ranks, m.rank_stride_words = len(block_ids), int(n_ranks), int(rank_stride)
        if getattr(self, "_apply_bar", None) is None:
            self._apply_bar = torch.zeros(2, dtype=torch.int32, device=gathered.device)
        _lib.check(_lib.lib.bliss_exp3_apply_ranks(C.byref(m), gathered.data_ptr(), self._apply_bar.data_ptr(), self._err.data_ptr(),
                                                   _stream()), "bliss_exp3_apply_ranks")

    def normalize(self, idx, g):
        """F.normalize(self.exp3_weights[idx], p=1, dim=0), bandit_sampler.py:249 (bit-exact, skipped on the
        device when the bf16 norm is 1.0)."""
        self._settle()
        _lib.check(_lib.lib.bliss_exp3_normalize(self._w_pos[idx].data_ptr(), g.num_edges(), self._row_sum[idx].data_ptr(),
                                                 self._scratch[idx].data_ptr(), self._norms[idx:].data_ptr(), _stream()),
                   "bliss_exp3_normalize")

    def check_errors(self):
        """Raise if any exp3 kernel flagged a non-finite weight (one sync; call off the hot path)."""
        bits = int(self._err.item()) | int((self._scratch[:, 0] >> 20).max().item())
        if bits:
            raise RuntimeError(f"exp3 kernel error 0x{bits:x}: {_lib.err_string(bits)}")


class PoissonBanditLadiesSampler(BanditLadiesSampler):
    """bandit_sampler.py:369-425."""

    _poisson = True
