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
"""
import contextlib
import sys
import types

import torch

NID = "_ID"
EID = "_ID"


# ------------------------------------------------------------------ exact sums
def _bf16_to_int(x, frac=150):
    bits = x.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    out = []
    for b in bits.tolist():
        s, e, m = b >> 15, (b >> 7) & 0xFF, b & 0x7F
        assert e != 255, "non-finite term"
        if e == 0:
            v = m << (frac - 133)
        else:
            v = (m | 0x80) << (e - 134 + frac)
        out.append(-v if s else v)
    return out


def _int_to_bf16_bits(n, frac=150):
    if n == 0:
        return 0
    s = 1 if n < 0 else 0
    n = abs(n)
    msb = n.bit_length() - 1
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
    assert 0 < e < 255
    return (s << 15) | (e << 7) | (q & 0x7F)


def _segment_sum(values, seg, nseg):
    """Exact per-segment sum of bf16 (or any float) values, rounded once to bf16."""
    values = values.bfloat16() if values.dtype != torch.bfloat16 else values
    acc = [0] * nseg
    for v, k in zip(_bf16_to_int(values), seg.tolist()):
        acc[k] += v
    bits = torch.tensor([_int_to_bf16_bits(a) for a in acc], dtype=torch.int64)
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
        try:
            yield
        finally:
            self.edata.clear(); self.edata.update(saved_e)
            self.srcdata.clear(); self.srcdata.update(saved_n)

    def update_all(self, msg, red):
        kind_m, a, b = msg
        kind_r, c, out = red
        assert kind_m == "copy_e" and kind_r == "sum"
        self.dstdata[out] = _segment_sum(self.edata[a], self._dst, self._nd)

    def apply_edges(self, fn):
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
    fn.sum = lambda a, b: ("sum", a, b)
    dgl.function = fn
    dl = types.ModuleType("dgl.dataloading")
    dl.BlockSampler = BlockSampler
    dgl.dataloading = dl
    sys.modules.update({"dgl": dgl, "dgl.ops": dgl.ops, "dgl.function": fn, "dgl.dataloading": dl})
    return dgl
