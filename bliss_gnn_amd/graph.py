"""Graph and Block (MFG) containers with the slice of the DGL surface the reference consumes.

The reference hands ``dgl.DGLGraph`` / DGLBlock objects between its DataLoader, samplers, models
and callback (SURVEY.md section 8b lists every attribute touched).  DGL is not available on this
platform, so these two small classes provide that surface -- and only that -- over plain device
tensors: a CSC graph (``train_lightning.py:373`` keeps only CSC) and a CSR-by-destination block.
"""
import contextlib

import torch

NID = "_ID"   # dgl.NID
EID = "_ID"   # dgl.EID


class Graph:
    """CSC message graph.  ``indptr`` int64 [V+1], ``indices`` int32 [E] (source of every in-edge,
    grouped by destination), ``eid`` int32 [E] = edge id of every CSC position (None: identity).

    ``ndata`` / ``edata`` hold per-node / per-EDGE-ID tensors exactly like DGL frames."""

    is_block = False

    def __init__(self, indptr, indices, eid=None, ndata=None, edata=None):
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32
        assert indptr.device == indices.device
        self.indptr = indptr.contiguous()
        self.indices = indices.contiguous()
        self.eid = None if eid is None else eid.to(torch.int32).contiguous()
        # plain dicts are copied; any other mapping (a DGL frame seen through as_graph) is kept LIVE
        self.ndata = dict(ndata or {}) if ndata is None or isinstance(ndata, dict) else ndata
        self.edata = dict(edata or {}) if edata is None or isinstance(edata, dict) else edata
        self.idtype = torch.int32
        self._inv_eid = None
        self._pos_cache = {}

    # -- DGL surface ------------------------------------------------------------------
    @property
    def device(self):
        return self.indptr.device

    def num_nodes(self):
        return self.indptr.numel() - 1

    number_of_nodes = num_nodes

    def num_edges(self):
        return self.indices.numel()

    number_of_edges = num_edges

    def in_degrees(self, v=None):
        deg = (self.indptr[1:] - self.indptr[:-1]).to(self.idtype)
        return deg if v is None else deg[v.long()]

    def int(self):
        return self

    def formats(self, fmts=None):
        return self

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        g = Graph(self.indptr.to(device), self.indices.to(device), None if self.eid is None else self.eid.to(device),
                  {k: v.to(device) for k, v in self.ndata.items()}, {k: v.to(device) for k, v in self.edata.items()})
        return g

    # -- edge-id <-> CSC-position plumbing --------------------------------------------
    def by_position(self, edge_tensor):
        """[.., E] tensor indexed by edge id -> indexed by CSC position (a view when eid is identity)."""
        if self.eid is None:
            return edge_tensor
        return edge_tensor[..., self.eid.long()]

    def by_edge_id(self, pos_tensor):
        if self.eid is None:
            return pos_tensor
        if self._inv_eid is None:
            inv = torch.empty_like(self.eid, dtype=torch.int64)
            inv[self.eid.long()] = torch.arange(self.eid.numel(), device=self.device)
            self._inv_eid = inv
        return pos_tensor[..., self._inv_eid]

    def edata_by_position(self, key):
        """Cached CSC-position-ordered copy of ``edata[key]`` (kernels read columns contiguously)."""
        t = self.edata[key]
        tag = (t.data_ptr(), t._version)
        hit = self._pos_cache.get(key)
        if hit is None or hit[0] != tag:
            self._pos_cache[key] = (tag, self.by_position(t).contiguous())
        return self._pos_cache[key][1]


class _FrameView:
    """A live, read-mostly view of somebody else's frame (``dgl_graph.ndata`` / ``.edata``): what the samplers and the lazy
    block frames need of a mapping, without copying its tensors."""

    def __init__(self, frame):
        self._f = frame

    def __getitem__(self, key):
        return self._f[key]

    def __setitem__(self, key, value):
        self._f[key] = value

    def __contains__(self, key):
        return key in self._f

    def keys(self):
        return self._f.keys()

    def items(self):
        return [(k, self._f[k]) for k in self._f.keys()]


def as_graph(g, cache=None):
    """``g`` as a bliss Graph.  A bliss Graph passes through; anything else is taken to be a DGLGraph-like object -- what
    ``dgl.dataloading.DataLoader(g, ...)`` hands to ``sampler.sample(g, ids)`` (train_lightning.py:396-408) and what the
    callback hands to ``sampler.exp3(mfgs, g)`` (:469-471) -- and is read through the four members it must expose:
    ``adj_tensors('csc') -> (indptr, indices, edge_ids)``, ``ndata``, ``edata``, ``num_nodes()``.  The CSC arrays are
    converted once (int64 indptr, int32 indices / edge ids, as train_lightning.py:340-342 makes them) and cached in
    ``cache`` (a dict owned by the caller, keyed by ``id(g)``); the frames stay live views."""
    if isinstance(g, Graph):
        return g
    key = id(g)
    if cache is not None and key in cache and cache[key][0] is g:
        return cache[key][1]
    if not all(hasattr(g, a) for a in ("adj_tensors", "ndata", "edata", "num_nodes")):
        raise TypeError("expected a bliss_gnn_amd.Graph or a DGLGraph-like object exposing adj_tensors('csc'), ndata, edata, "
                        "num_nodes(); got %r" % type(g).__name__)
    indptr, indices, eids = g.adj_tensors("csc")
    if int(indptr.numel()) != int(g.num_nodes()) + 1:
        raise ValueError("adj_tensors('csc') returned an indptr that does not match num_nodes()")
    n = int(indices.numel())
    identity = eids is None or eids.numel() == 0 or bool((eids == torch.arange(n, device=eids.device, dtype=eids.dtype)).all())
    bg = Graph(indptr.to(torch.int64), indices.to(torch.int32), None if identity else eids.to(torch.int32),
               ndata=_FrameView(g.ndata), edata=_FrameView(g.edata))
    if cache is not None:
        cache[key] = (g, bg)
    return bg


class _LazyFrame(dict):
    """A frame that materialises parent features on first access, like DGL's lazy feature slicing:
    ``blocks[0].srcdata['features'] == g.ndata['features'][input_nodes]`` (train_lightning.py:138)."""

    def __init__(self, parent_frame, index_fn):
        super().__init__()
        self._parent = parent_frame
        self._index_fn = index_fn
        self._row_norms = []            # (gathered tensor, its bf16 row norms) for rows gathered by the fused kernel

    def __missing__(self, key):
        if self._parent is not None and key in self._parent:
            src, idx = self._parent[key], self._index_fn()
            if (src.is_cuda and src.dtype == torch.bfloat16 and src.dim() == 2 and src.stride(1) == 1 and src.shape[1] <= 6144
                    and idx.dtype == torch.int32 and idx.is_contiguous()):
                # bf16 feature rows: one HBM-bound pass gathers them AND leaves their row norms (the embed_norm the model
                # takes of blocks[0]'s input right afterwards, model.py:318-320)
                from . import _lib
                v = torch.empty(idx.numel(), src.shape[1], dtype=torch.bfloat16, device=src.device)
                nrm = torch.empty(idx.numel(), dtype=torch.bfloat16, device=src.device)
                _lib.check(_lib.lib.bliss_gather_rows(src.data_ptr(), src.stride(0), idx.data_ptr(), idx.numel(), src.shape[1],
                                                      v.data_ptr(), v.stride(0), nrm.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "bliss_gather_rows")
                self._row_norms.append((v, nrm))
            else:
                v = torch.index_select(src, 0, idx)                             # int32 ids are fine: no widening pass
            self[key] = v
            return v
        raise KeyError(key)

    def lazy(self, key):
        """``frame[key]`` as a not-yet-gathered (table, ids) pair when the frame has not materialised it and the rows are
        bf16 features (nn.LazyRows: the fused SAGE transform gathers them as its operand load); else the tensor itself."""
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        if self._parent is not None and key in self._parent:
            src = self._parent[key]
            idx = self._index_fn()
            if src.is_cuda and src.dtype == torch.bfloat16 and src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32 and idx.is_contiguous():
                from .nn import LazyRows
                return LazyRows(src, idx, self, key)
        return self[key]

    def row_norm_of(self, t):
        """bf16 row norms of ``t`` if ``t`` is a tensor this frame gathered with the fused kernel, else None."""
        for v, nrm in self._row_norms:
            if v is t:
                return nrm
        return None

    def __contains__(self, key):
        return dict.__contains__(self, key) or (self._parent is not None and key in self._parent)

    def update(self, other=(), **kw):   # dict.update bypasses __setitem__ semantics we rely on; keep simple
        for k, v in dict(other, **kw).items():
            self[k] = v


class Block:
    """One message-flow graph: ``n_src`` source nodes, the first ``n_dst`` of which are the
    destinations; edges stored CSR-by-destination in frontier order (``indptr``, ``src``) with the
    per-edge ``dst`` alongside.  Mirrors the DGLBlock members listed in SURVEY.md section 8b."""

    is_block = True

    def __init__(self, g, n_src, n_dst, indptr, src, dst, pos, eid, src_nid):
        self.g = g
        self._n_src, self._n_dst = int(n_src), int(n_dst)
        self.indptr, self.src, self.dst, self.pos = indptr, src, dst, pos
        self.idtype = torch.int32
        self.srcdata = _LazyFrame(g.ndata if g is not None else None, lambda: self.srcdata[NID])
        self.dstdata = _LazyFrame(g.ndata if g is not None else None, lambda: self.dstdata[NID])
        self.edata = _LazyFrame(g.edata if g is not None else None, lambda: self.edata[EID])
        self.srcdata[NID] = src_nid
        self.dstdata[NID] = src_nid[: self._n_dst]
        self.edata[EID] = eid
        self._transposed = None
        self._nnz_ptr = 0            # device address of the true edge count when the arrays are capacity-padded

    @property
    def device(self):
        return self.src.device

    def num_src_nodes(self):
        return self._n_src

    def num_dst_nodes(self):
        return self._n_dst

    number_of_dst_nodes = num_dst_nodes
    number_of_src_nodes = num_src_nodes

    def num_edges(self):
        return self.src.numel()

    number_of_edges = num_edges

    def in_degrees(self):
        return (self.indptr[1:] - self.indptr[:-1]).to(self.idtype)

    def edges(self):
        return self.src, self.dst

    def int(self):
        return self

    def to(self, device):
        assert torch.device(device) == self.device, "sampling device must equal training device (SURVEY.md 3.1)"
        return self

    @contextlib.contextmanager
    def local_scope(self):
        saved = [dict.copy(f) for f in (self.srcdata, self.dstdata, self.edata)]
        try:
            yield
        finally:
            for f, s in zip((self.srcdata, self.dstdata, self.edata), saved):
                dict.clear(f)
                dict.update(f, s)

    # -- the message-passing calls the reference's layers make on a block (model.py:82, :98) ------
    def apply_edges(self, func):
        """``graph.apply_edges(fn.u_add_v('el', 'er', 'e'))`` (model.py:82): materialises the per-edge tensor, as DGL does
        (bliss_gnn_amd.nn.GATv2Conv fuses this step away; this entry exists so that layer code written against DGL runs)."""
        if getattr(func, "kind", None) != "u_add_v":
            raise NotImplementedError("apply_edges: only fn.u_add_v is used by the reference (model.py:82)")
        self.edata[func.out] = self.srcdata[func.lhs][self.src.long()] + self.dstdata[func.rhs][self.dst.long()]

    def update_all(self, message_func, reduce_func):
        """``graph.update_all(fn.u_mul_e(h, w, 'm') | fn.copy_u(h, 'm'), fn.sum('m', out) | fn.mean('m', out))`` on the SpMM
        kernels (csrc/spmm.hip, csrc/gat.hip); the result lands in ``dstdata[out]`` (model.py:98-99)."""
        from . import nn as bnn
        kind, red = getattr(message_func, "kind", None), getattr(reduce_func, "kind", None)
        if kind not in ("u_mul_e", "copy_u") or red not in ("sum", "mean") or reduce_func.msg != message_func.out:
            raise NotImplementedError("update_all: (u_mul_e | copy_u) with (sum | mean), as the reference's layers use them")
        h = self.srcdata[message_func.lhs]
        w = self.edata[message_func.rhs] if kind == "u_mul_e" else None
        S = self.num_dst_nodes()
        if h.dim() == 3 and w is not None and w.dim() == 3:                        # GAT: [K,H,D] x [B,H,1]  (model.py:98)
            if red != "sum":
                raise NotImplementedError("per-head messages are summed in the reference (model.py:98)")
            K, H, D = h.shape
            out = bnn._GatAggregate.apply(w.reshape(-1, H), h.reshape(K, H * D), self, H, D).view(S, H, D)
        else:
            h2 = h.reshape(h.shape[0], -1)
            out = bnn.weighted_aggregate(self, h2, None if w is None else w.reshape(-1), mean=(red == "mean"))
            out = out.view((S,) + tuple(h.shape[1:]))
        self.dstdata[reduce_func.out] = out

    def transposed(self):
        """Edges grouped by SOURCE (stable: ascending edge index inside a source): ``(t_indptr int32
        [n_src+1], t_edge int32 [len(src)])``.  Only the SpMM backward needs it; built on the device
        from the (possibly capacity-padded) edge arrays and the device-resident edge count."""
        if self._transposed is None:
            import ctypes as C
            from . import _lib
            n_src, cap_b = self._n_src, int(self.src.numel())
            t_indptr = torch.empty(n_src + 1, dtype=torch.int32, device=self.device)
            t_edge = torch.empty(max(cap_b, 1), dtype=torch.int32, device=self.device)
            nbytes = int(_lib.lib.bliss_block_transpose_temp_bytes(cap_b, n_src)) if cap_b else 0
            if nbytes < 0:
                raise RuntimeError("bliss_block_transpose_temp_bytes failed")
            temp = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib.bliss_block_transpose(self.src.data_ptr(), self._nnz_ptr, cap_b, cap_b, n_src, t_indptr.data_ptr(),
                                                      t_edge.data_ptr(), temp.data_ptr(), nbytes,
                                                      torch.cuda.current_stream().cuda_stream), "bliss_block_transpose")
            self._transposed = (t_indptr, t_edge)
        return self._transposed


def full_neighbor_block(g, b0, b1):
    """The block ``MultiLayerFullNeighborSampler(1)`` yields for the contiguous seed range b0..b1-1 (the reference's
    ``inference`` loaders, model.py:248-263, 347-362, 453-468): ALL in-edges of the seeds, destinations first among the
    sources, edge order = CSC order.  (The extra sources are numbered in ascending id order instead of first appearance;
    no destination row depends on that numbering.)  Evaluation-time helper: it synchronises once for the source count."""
    S = b1 - b0
    e0, e1 = int(g.indptr[b0]), int(g.indptr[b1])
    indptr = (g.indptr[b0:b1 + 1] - e0).to(torch.int32)
    col = g.indices[e0:e1].long()
    outside = (col < b0) | (col >= b1)
    others = torch.unique(col[outside])
    src = torch.where(outside, S + torch.searchsorted(others, col), col - b0).to(torch.int32)
    deg = (indptr[1:] - indptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(S, device=col.device, dtype=torch.int32), deg, output_size=e1 - e0)
    src_nid = torch.cat([torch.arange(b0, b1, device=col.device), others]).to(torch.int32)
    pos = torch.arange(e0, e1, device=col.device, dtype=torch.int32) if e1 < 2 ** 31 else None
    eid = g.eid[e0:e1] if g.eid is not None else pos
    return Block(g, src_nid.numel(), S, indptr, src, dst, pos, eid, src_nid)
