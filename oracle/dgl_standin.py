"""A stand-in for the handful of ``dgl`` entry points the reference samplers call.

TEST INFRASTRUCTURE ONLY, and only for ``tests/golden/make_golden.py`` (run in the
build container, where /root/reference exists).  It lets the reference's
UNMODIFIED ``bandit_sampler.py`` / ``ladies_sampler.py`` execute -- their Python
control flow and every torch call are the reference's own -- on top of graph
primitives written here from DGL's documented behaviour ([DGL-recalled], see
SURVEY.md section 8c).  It is deliberately an independent, slow, obviously-correct
implementation (COO edge lists, Python-int exact sums) so that agreement with
``oracle/bliss_oracle.py`` means something.

Surface (call sites in /root/reference): dgl.NID/EID, dgl.dataloading.BlockSampler,
dgl.in_subgraph (bandit_sampler.py:123), dgl.compact_graphs (:125), dgl.reverse (:69),
dgl.edge_subgraph (:298), dgl.to_block (:322), dgl.ops.{copy_e_sum,e_div_v,e_div_u,
v_add_e,e_mul_v,u_div_e,e_dot_v} (:67-73,:129-137,:150-154,:186-189,:240-242,:314-320),
dgl.function.{copy_e,sum} + update_all/apply_edges/local_scope (:21-27),
DGLGraph.{subgraph,edges,in_degrees,out_degrees,num_nodes,num_edges,idtype,device,
ndata,edata,srcdata,dstdata}.

Round 3, for the MODEL side (/root/reference/model.py run unmodified by make_golden.py): dgl.nn.GATv2Conv as the base
class ``custom_GATv2Conv`` derives from (model.py:13; only its constructor state is used: fc_src, fc_dst, attn, feat_drop,
attn_drop, leaky_relu, res_fc, activation, share_weights -- the forward is the reference's own, model.py:48-112),
dgl.nn.SAGEConv('mean') (model.py:303-308), dgl.nn.functional.edge_softmax (:89), dgl.function.{u_add_v,u_mul_e,copy_u,
sum,mean} with DGLBlock.apply_edges / update_all (:82, :98), dgl.base.DGLError (:10).  [DGL-recalled] numerics of those
primitives, stated where they are implemented below.
"""
import contextlib
import sys
import types

import torch

NID = "_ID"
EID = "_ID"


# ------------------------------------------------------------------ exact sums
def _bf16_to_int(x, frac=150):
    """bf16 -> Python int scaled by 2**frac (exact, subnormals included); None for a non-finite term."""
    bits = x.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    out = []
    for b in bits.tolist():
        s, e, m = b >> 15, (b >> 7) & 0xFF, b & 0x7F
        if e == 255:
            out.append(None)
            continue
        if e == 0:
            v = m << (frac - 133)
        else:
            v = (m | 0x80) << (e - 134 + frac)
        out.append(-v if s else v)
    return out


def _int_to_bf16_bits(n, frac=150):
    """exact Python int * 2**-frac -> bf16 bits, round to nearest even; subnormal results and overflow to inf like IEEE."""
    if n == 0:
        return 0
    s = 1 if n < 0 else 0
    n = abs(n)
    msb = n.bit_length() - 1
    if msb - frac + 127 <= 0:                      # below the normal range: units of 2**-133 (the subnormal spacing)
        sh = frac - 133
        q, rem, half = n >> sh, n & ((1 << sh) - 1), 1 << (sh - 1)
        if rem > half or (rem == half and (q & 1)):
            q += 1
        return (s << 15) | q                       # q == 128 is the smallest normal: same encoding
    if msb > 7:
        sh = msb - 7
        q, rem, half = n >> sh, n & ((1 << sh) - 1), 1 << (sh - 1)
        if rem > half or (rem == half and (q & 1)):
            q += 1
    else:
        q = n << (7 - msb)
    e = msb - frac + 127
    if q >= 256:
        q >>= 1
        e += 1
    if e >= 255:
        return (s << 15) | 0x7F80
    return (s << 15) | (e << 7) | (q & 0x7F)


def _segment_sum(values, seg, nseg):
    """Exact per-segment sum of bf16 (or any float) values, rounded once to bf16; non-finite terms follow IEEE addition
    (NaN, or +inf and -inf together -> NaN; else the infinity)."""
    values = values.bfloat16() if values.dtype != torch.bfloat16 else values
    acc = [0] * nseg
    special = {}
    for v, k, raw in zip(_bf16_to_int(values), seg.tolist(), values.float().tolist()):
        if v is None:
            special.setdefault(k, []).append(raw)
        else:
            acc[k] += v
    bits = [_int_to_bf16_bits(a) for a in acc]
    for k, raws in special.items():
        nan = any(r != r for r in raws) or (any(r > 0 for r in raws) and any(r < 0 for r in raws))
        bits[k] = 0x7FC0 if nan else (0x7F80 if raws[0] > 0 else 0xFF80)
    bits = torch.tensor(bits, dtype=torch.int64)
    bits = torch.where(bits >= 32768, bits - 65536, bits).to(torch.int16)
    return bits.view(torch.bfloat16)


# ------------------------------------------------------------------ graph
class Graph:
    def __init__(self, src, dst, num_src, num_dst=None, idtype=torch.int32, is_block=False):
        self._src = src.to(torch.int64)
        self._dst = dst.to(torch.int64)
        self._ns = int(num_src)
        self._nd = int(num_src if num_dst is None else num_dst)
        self.idtype = idtype
        self.device = torch.device("cpu")
        self.is_block = is_block
        self.edata = {}
        if is_block:
            self.srcdata, self.dstdata = {}, {}
        else:
            self.ndata = {}
            self.srcdata = self.dstdata = self.ndata

    # sizes
    def num_nodes(self):
        assert not self.is_block
        return self._ns

    number_of_nodes = num_nodes

    def num_src_nodes(self):
        return self._ns

    def num_dst_nodes(self):
        return self._nd

    number_of_dst_nodes = num_dst_nodes

    def num_edges(self):
        return self._src.numel()

    number_of_edges = num_edges

    def edges(self):
        return self._src.to(self.idtype), self._dst.to(self.idtype)

    def all_edges(self):
        return self.edges()

    def in_degrees(self, v=None):
        deg = torch.zeros(self._nd, dtype=torch.int64).index_add_(0, self._dst, torch.ones_like(self._dst))
        deg = deg.to(self.idtype)
        return deg if v is None else deg[v.long()]

    def out_degrees(self, v=None):
        deg = torch.zeros(self._ns, dtype=torch.int64).index_add_(0, self._src, torch.ones_like(self._src))
        deg = deg.to(self.idtype)
        return deg if v is None else deg[v.long()]

    def int(self):
        return self

    def to(self, device):
        return self

    # structure ops
    def subgraph(self, nodes):
        """node-induced subgraph: nodes renumbered in the given order, edge order kept."""
        nodes = nodes.to(torch.int64)
        new = torch.full((self._ns,), -1, dtype=torch.int64)
        new[nodes] = torch.arange(nodes.numel())
        keep = (new[self._src] >= 0) & (new[self._dst] >= 0)
        sg = Graph(new[self._src[keep]], new[self._dst[keep]], nodes.numel(), idtype=self.idtype)
        sg.ndata[NID] = nodes.to(self.idtype)
        sg.edata[EID] = torch.nonzero(keep, as_tuple=True)[0].to(self.idtype)
        _inherit(self, sg, node_idx=nodes, edge_idx=torch.nonzero(keep, as_tuple=True)[0])
        return sg

    @contextlib.contextmanager
    def local_scope(self):
        saved_e, saved_n = dict(self.edata), dict(self.srcdata)
        saved_d = dict(self.dstdata) if self.dstdata is not self.srcdata else None
        try:
            yield
        finally:
            self.edata.clear(); self.edata.update(saved_e)
            self.srcdata.clear(); self.srcdata.update(saved_n)
            if saved_d is not None:
                self.dstdata.clear(); self.dstdata.update(saved_d)

    def update_all(self, msg, red):
        """[DGL-recalled] g-SpMM.  copy_e: the edge values; u_mul_e / copy_u: message = src row (x edge value), computed
        and accumulated in fp32 from the stored (bf16) operands in edge order of each destination, ONE rounding to the
        operand dtype at the end (DGL's kernels do not materialise the message tensor; their accumulation type for bf16 is
        not knowable offline -- see oracle/bliss_oracle.py header).  sum / mean."""
        kind_m, a, b = msg
        kind_r, c, out = red
        if kind_m == "copy_e":
            assert kind_r == "sum"
            self.dstdata[out] = _segment_sum(self.edata[a], self._dst, self._nd)
            return
        assert kind_m in ("u_mul_e", "copy_u") and kind_r in ("sum", "mean")
        x = self.srcdata[a]
        m = x.float()[self._src]
        if kind_m == "u_mul_e":
            e = self.edata[b].float()
            while e.dim() < m.dim():
                e = e.unsqueeze(-1)
            m = m * e
        acc = torch.zeros((self._nd,) + tuple(m.shape[1:]), dtype=torch.float32).index_add_(0, self._dst, m)
        acc = acc.to(x.dtype)
        if kind_r == "mean":
            # [DGL-recalled] dgl/ops/spmm.py:gspmm: 'mean' = the 'sum' result (already in the operand dtype) divided by
            # clamp(in_degrees, 1) cast to that dtype -- two roundings for bf16
            deg = torch.zeros(self._nd, dtype=torch.float32).index_add_(0, self._dst, torch.ones(self._dst.numel()))
            acc = acc / deg.clamp(min=1).to(x.dtype).view((-1,) + (1,) * (acc.dim() - 1))
        self.dstdata[out] = acc

    def apply_edges(self, fn):
        if isinstance(fn, tuple):                      # built-in message function: element-wise in the operand dtype
            kind, a, b, out = fn
            assert kind == "u_add_v"
            self.edata[out] = self.srcdata[a][self._src] + self.dstdata[b][self._dst]
            return
        edges = types.SimpleNamespace(dst={k: v[self._dst] for k, v in self.dstdata.items() if v.shape[0] == self._nd},
                                      src={k: v[self._src] for k, v in self.srcdata.items() if v.shape[0] == self._ns},
                                      data=self.edata)
        self.edata.update(fn(edges))


def _inherit(parent, child, node_idx=None, edge_idx=None):
    """DGL's lazy feature slicing: sub-structures carry the parent's features."""
    if edge_idx is not None:
        for k, v in parent.edata.items():
            if k != EID and v.shape[0] == parent.num_edges():
                child.edata.setdefault(k, v[edge_idx])
    if node_idx is not None and not parent.is_block and not child.is_block:
        for k, v in parent.ndata.items():
            if k != NID and v.shape[0] == parent._ns:
                child.ndata.setdefault(k, v[node_idx])


def graph_from_csc(indptr, indices, eid=None, idtype=torch.int32):
    """Build the message graph g from CSC; edge i of the COO list is the edge with id i."""
    indptr = indptr.to(torch.int64)
    V = indptr.numel() - 1
    dst_pos = torch.repeat_interleave(torch.arange(V), indptr[1:] - indptr[:-1])
    src_pos = indices.to(torch.int64)
    if eid is None:
        src, dst = src_pos, dst_pos
    else:
        src = torch.empty_like(src_pos); dst = torch.empty_like(dst_pos)
        src[eid.long()] = src_pos; dst[eid.long()] = dst_pos
    g = Graph(src, dst, V, idtype=idtype)
    g._csc = (indptr, indices.to(torch.int64), None if eid is None else eid.to(torch.int64))
    return g


def in_subgraph(g, seeds):
    """All in-edges of ``seeds``: grouped by seed in seed order, CSC order inside a seed."""
    indptr, indices, eid = g._csc
    pos = torch.cat([torch.arange(indptr[s], indptr[s + 1]) for s in seeds.tolist()]) if len(seeds) else torch.zeros(0, dtype=torch.int64)
    dst = torch.cat([torch.full((int(indptr[s + 1] - indptr[s]),), s, dtype=torch.int64) for s in seeds.tolist()]) if len(seeds) else pos
    sg = Graph(indices[pos], dst, g.num_nodes(), idtype=g.idtype)
    e = pos if eid is None else eid[pos]
    sg.edata[EID] = e.to(g.idtype)
    _inherit(g, sg, edge_idx=e)
    for k, v in g.ndata.items():
        sg.ndata[k] = v
    return sg


def compact_graphs(sg, always_preserve):
    """Drop isolated nodes; ``always_preserve`` first, then first appearance (src list, then dst list)."""
    order, seen = [], set()
    for n in always_preserve.tolist() + sg._src.tolist() + sg._dst.tolist():
        if n not in seen:
            seen.add(n); order.append(n)
    nid = torch.tensor(order, dtype=torch.int64)
    new = torch.full((sg._ns,), -1, dtype=torch.int64)
    new[nid] = torch.arange(nid.numel())
    cg = Graph(new[sg._src], new[sg._dst], nid.numel(), idtype=sg.idtype)
    cg.edata.update(sg.edata)
    for k, v in sg.ndata.items():
        if k != NID:
            cg.ndata[k] = v[nid]
    cg.ndata[NID] = nid.to(sg.idtype)
    return cg


def reverse(g, copy_edata=False):
    r = Graph(g._dst, g._src, g._ns, idtype=g.idtype)
    r.ndata.update(g.ndata)
    if copy_edata:
        r.edata.update(g.edata)
    return r


def edge_subgraph(g, mask, relabel_nodes=True):
    assert relabel_nodes is False
    idx = torch.nonzero(mask, as_tuple=True)[0] if mask.dtype == torch.bool else mask.long()
    eg = Graph(g._src[idx], g._dst[idx], g._ns, idtype=g.idtype)
    eg.edata[EID] = idx.to(g.idtype)
    _inherit(g, eg, edge_idx=idx)
    for k, v in g.ndata.items():
        if k != NID:
            eg.ndata[k] = v
    return eg


def to_block(g, dst_nodes):
    """dst nodes first (given order), then the other sources by first appearance; edge order kept."""
    dst_nodes = dst_nodes.to(torch.int64)
    order, seen = dst_nodes.tolist(), set(dst_nodes.tolist())
    for n in g._src.tolist():
        if n not in seen:
            seen.add(n); order.append(n)
    src_ids = torch.tensor(order, dtype=torch.int64)
    new_s = torch.full((g._ns,), -1, dtype=torch.int64); new_s[src_ids] = torch.arange(src_ids.numel())
    new_d = torch.full((g._ns,), -1, dtype=torch.int64); new_d[dst_nodes] = torch.arange(dst_nodes.numel())
    assert bool((new_d[g._dst] >= 0).all())
    b = Graph(new_s[g._src], new_d[g._dst], src_ids.numel(), dst_nodes.numel(), idtype=g.idtype, is_block=True)
    b.srcdata[NID] = src_ids.to(g.idtype)
    b.dstdata[NID] = dst_nodes.to(g.idtype)
    b.edata[EID] = torch.arange(g.num_edges()).to(g.idtype)
    for k, v in g.edata.items():
        if k != EID:
            b.edata[k] = v
    for k, v in g.ndata.items():
        if k != NID:
            b.srcdata[k] = v[src_ids]; b.dstdata[k] = v[dst_nodes]
    return b


# ------------------------------------------------------------------ dgl.ops
def _ops():
    m = types.ModuleType("dgl.ops")
    m.copy_e_sum = lambda g, e: _segment_sum(e, g._dst, g._nd)
    m.e_div_v = lambda g, e, v: e / v[g._dst]
    m.e_div_u = lambda g, e, u: e / u[g._src]
    m.e_mul_v = lambda g, e, v: e * v[g._dst]
    m.v_add_e = lambda g, v, e: v[g._dst] + e
    m.u_div_e = lambda g, u, e: u[g._src] / e
    m.e_dot_v = lambda g, e, v: e * v[g._dst]
    return m


# ------------------------------------------------------------------ dgl.nn (constructor state only; [DGL-recalled])
def edge_softmax(graph, e):
    """[DGL-recalled] dgl.nn.functional.edge_softmax (dgl/backend/pytorch/sparse.py: EdgeSoftmax.forward): four ops, each
    in the dtype of ``e``:  max over the in-edges of a destination; exp(e - max); sum over the in-edges (exact sum of the
    rounded terms, rounded once: the stand-in's copy_e_sum contract); out = score / sum."""
    shape = e.shape
    x = e.reshape(shape[0], -1)
    nd, dst = graph._nd, graph._dst
    mx = torch.full((nd, x.shape[1]), -float("inf"), dtype=torch.float32)
    mx = mx.scatter_reduce(0, dst[:, None].expand(-1, x.shape[1]), x.float(), "amax").to(e.dtype)
    score = torch.exp(x - mx[dst])
    ssum = torch.stack([_segment_sum(score[:, h].contiguous(), dst, nd) for h in range(x.shape[1])], 1) if e.dtype == torch.bfloat16 \
        else torch.zeros(nd, x.shape[1], dtype=e.dtype).index_add_(0, dst, score)
    return (score / ssum[dst]).reshape(shape)


def _nn():
    import torch.nn as tnn

    class GATv2Conv(tnn.Module):
        """[DGL-recalled] dgl.nn.GATv2Conv.__init__ / reset_parameters (dgl 2.2.1): field names as the reference's forward reads them."""

        def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0, negative_slope=0.2, residual=False,
                     activation=None, allow_zero_in_degree=False, bias=True, share_weights=False):
            super().__init__()
            self._num_heads = num_heads
            self._in_src_feats = self._in_dst_feats = in_feats
            self._out_feats = out_feats
            self._allow_zero_in_degree = allow_zero_in_degree
            self.fc_src = tnn.Linear(in_feats, out_feats * num_heads, bias=bias)
            self.fc_dst = self.fc_src if share_weights else tnn.Linear(in_feats, out_feats * num_heads, bias=bias)
            self.attn = tnn.Parameter(torch.empty(1, num_heads, out_feats))
            self.feat_drop = tnn.Dropout(feat_drop)
            self.attn_drop = tnn.Dropout(attn_drop)
            self.leaky_relu = tnn.LeakyReLU(negative_slope)
            if residual:
                self.res_fc = tnn.Linear(in_feats, num_heads * out_feats, bias=bias) if in_feats != out_feats * num_heads else tnn.Identity()
            else:
                self.register_buffer("res_fc", None)
            self.activation = activation
            self.share_weights = share_weights
            self.bias = bias
            gain = tnn.init.calculate_gain("relu")
            tnn.init.xavier_normal_(self.fc_src.weight, gain=gain)
            if bias:
                tnn.init.constant_(self.fc_src.bias, 0)
            if not share_weights:
                tnn.init.xavier_normal_(self.fc_dst.weight, gain=gain)
                if bias:
                    tnn.init.constant_(self.fc_dst.bias, 0)
            tnn.init.xavier_normal_(self.attn, gain=gain)
            if isinstance(self.res_fc, tnn.Linear):
                tnn.init.xavier_normal_(self.res_fc.weight, gain=gain)
                if bias:
                    tnn.init.constant_(self.res_fc.bias, 0)

    class SAGEConv(tnn.Module):
        """[DGL-recalled] dgl.nn.SAGEConv(in, out, 'mean').forward(graph, feat, edge_weight): fc_neigh (bias-free) BEFORE the
        aggregation iff in > out; update_all(u_mul_e('h','_edge_weight','m'), mean('m','neigh')); rst = fc_self(h_dst) + h_neigh."""

        def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True, norm=None, activation=None):
            super().__init__()
            assert aggregator_type == "mean"
            self._in_src_feats = self._in_dst_feats = in_feats
            self._out_feats = out_feats
            self.fc_neigh = tnn.Linear(in_feats, out_feats, bias=False)
            self.fc_self = tnn.Linear(in_feats, out_feats, bias=bias)
            gain = tnn.init.calculate_gain("relu")
            tnn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
            tnn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

        def forward(self, graph, feat, edge_weight=None):
            with graph.local_scope():
                feat_src = feat
                feat_dst = feat[: graph.number_of_dst_nodes()]
                msg = ("copy_u", "h", "m")
                if edge_weight is not None:
                    graph.edata["_edge_weight"] = edge_weight
                    msg = ("u_mul_e", "h", "_edge_weight", "m")
                lin_before_mp = self._in_src_feats > self._out_feats
                graph.srcdata["h"] = self.fc_neigh(feat_src) if lin_before_mp else feat_src
                graph.update_all(msg[:3], ("mean", "m", "neigh"))
                h_neigh = graph.dstdata["neigh"]
                if not lin_before_mp:
                    h_neigh = self.fc_neigh(h_neigh)
                return self.fc_self(feat_dst) + h_neigh

    m = types.ModuleType("dgl.nn")
    m.GATv2Conv, m.SAGEConv = GATv2Conv, SAGEConv
    m.GraphConv = None                                     # model.py:397-417 only runs if a GCN is built (never from the CLI)
    f = types.ModuleType("dgl.nn.functional")
    f.edge_softmax = edge_softmax
    m.functional = f
    return m, f


class DGLError(Exception):
    pass


class BlockSampler:
    def __init__(self, *a, **k):
        pass

    def sample(self, g, seed_nodes, exclude_eids=None):
        return self.sample_blocks(g, seed_nodes, exclude_eids=exclude_eids)


def install():
    """Register the stand-in as ``dgl`` in sys.modules (only if the real one is absent)."""
    if "dgl" in sys.modules and not getattr(sys.modules["dgl"], "__standin__", False):
        raise RuntimeError("a real dgl is importable; the stand-in must not shadow it")
    dgl = types.ModuleType("dgl")
    dgl.__standin__ = True
    dgl.NID, dgl.EID = NID, EID
    dgl.in_subgraph, dgl.compact_graphs, dgl.reverse = in_subgraph, compact_graphs, reverse
    dgl.edge_subgraph, dgl.to_block = edge_subgraph, to_block
    dgl.ops = _ops()
    fn = types.ModuleType("dgl.function")
    fn.copy_e = lambda a, b: ("copy_e", a, b)
    fn.copy_u = lambda a, b: ("copy_u", a, b)
    fn.u_mul_e = lambda a, b, c: ("u_mul_e", a, b)
    fn.u_add_v = lambda a, b, c: ("u_add_v", a, b, c)
    fn.sum = lambda a, b: ("sum", a, b)
    fn.mean = lambda a, b: ("mean", a, b)
    dgl.function = fn
    dgl.nn, nnf = _nn()
    base = types.ModuleType("dgl.base")
    base.DGLError = DGLError
    dgl.base = base
    dl = types.ModuleType("dgl.dataloading")
    dl.BlockSampler = BlockSampler
    dgl.dataloading = dl
    sys.modules.update({"dgl": dgl, "dgl.ops": dgl.ops, "dgl.function": fn, "dgl.dataloading": dl, "dgl.nn": dgl.nn,
                        "dgl.nn.functional": nnf, "dgl.base": base})
    return dgl
