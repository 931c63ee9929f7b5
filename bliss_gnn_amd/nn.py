"""Message-passing layers over bliss Blocks: the ``dglnn.SAGEConv`` the reference instantiates
(model.py:303-308) with its g-SpMM replaced by the gfx950 kernels of csrc/spmm.hip.

[DGL-recalled] dglnn.SAGEConv(in, out, 'mean'):  rst = fc_self(h[:S]) + fc_neigh o mean_w(h);
fc_neigh (bias-free) is applied BEFORE aggregation iff in > out; fc_self carries the bias; both
weights xavier-uniform with gain('relu').  The dense transforms are plain library GEMMs (hipBLASLt
through torch.nn.functional.linear); the aggregation and its backward are ours.
"""
import torch
import torch.nn as nn

from . import _lib
from ._engine import _stream


def embed_norm(h):
    """``th.reshape(th.norm(h, dim=1, keepdim=True), (-1,))`` of model.py:318-320 -> bf16 [K]."""
    h = h.detach()
    if h.dtype != torch.bfloat16:
        h = h.bfloat16()
    if h.stride(1) != 1:
        h = h.contiguous()
    out = torch.empty(h.shape[0], dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.lib.bliss_embed_norm(h.data_ptr(), h.shape[0], h.shape[1], h.stride(0), out.data_ptr(), _stream()),
               "bliss_embed_norm")
    return out


class _WeightedAggregate(torch.autograd.Function):
    """out[i] = (1/deg_i if mean) * sum_{e -> i} w_e h[src_e]; gradient w.r.t. h only (the sampler's
    edge weights carry no grad)."""

    @staticmethod
    def forward(ctx, h, block, edge_weight, mean, out_fp32):
        assert h.is_cuda and h.dtype == torch.bfloat16, "bf16 features on the GPU (load_graph.py:7)"
        if h.stride(1) != 1:
            h = h.contiguous()
        S, D = block.num_dst_nodes(), h.shape[1]
        out = torch.empty(S, D, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=h.device)
        w = None
        if edge_weight is not None:
            w = edge_weight.reshape(-1)
            if w.dtype != torch.bfloat16:
                w = w.bfloat16()
            w = w.contiguous()
            assert w.numel() == block.num_edges()
        B = block.num_edges()
        part = torch.empty(2 * ((B + 63) // 64) * D, dtype=torch.float32, device=h.device) if B > 0 else None
        _lib.check(_lib.lib.bliss_spmm_fwd(block.indptr.data_ptr(), block.src.data_ptr(), block.dst.data_ptr(),
                                           0 if w is None else w.data_ptr(), h.data_ptr(), h.stride(0), S, block._nnz_ptr, B, D,
                                           int(mean),
                                           out.data_ptr(), out.stride(0), int(out_fp32), 0 if part is None else part.data_ptr(),
                                           _stream()), "bliss_spmm_fwd")
        ctx.block, ctx.w, ctx.mean, ctx.n_src = block, w, mean, h.shape[0]
        return out

    @staticmethod
    def backward(ctx, gout):
        block, w = ctx.block, ctx.w
        gout = gout.contiguous()
        if gout.dtype != torch.bfloat16:
            gout = gout.bfloat16()
        D = gout.shape[1]
        t_indptr, t_edge = block.transposed()
        gh = torch.empty(ctx.n_src, D, dtype=torch.bfloat16, device=gout.device)
        B = block.num_edges()
        part = torch.empty(2 * ((B + 63) // 64) * D, dtype=torch.float32, device=gout.device) if B > 0 else None
        _lib.check(_lib.lib.bliss_spmm_bwd(t_indptr.data_ptr(), t_edge.data_ptr(), block.src.data_ptr(), block.dst.data_ptr(),
                                           block.indptr.data_ptr(), 0 if w is None else w.data_ptr(), gout.data_ptr(),
                                           gout.stride(0), ctx.n_src, block._nnz_ptr, B, D, int(ctx.mean), gh.data_ptr(),
                                           gh.stride(0), 0,
                                           0 if part is None else part.data_ptr(), _stream()), "bliss_spmm_bwd")
        return gh, None, None, None, None


def weighted_aggregate(block, h, edge_weight=None, mean=True, out_fp32=False):
    return _WeightedAggregate.apply(h, block, edge_weight, mean, out_fp32)


class SAGEConv(nn.Module):
    """dglnn.SAGEConv(in_feats, out_feats, 'mean') as used at model.py:303-308, 321-329."""

    def __init__(self, in_feats, out_feats, aggregator_type="mean", feat_drop=0.0, bias=True, norm=None, activation=None):
        super().__init__()
        if aggregator_type != "mean":
            raise NotImplementedError("the reference only builds SAGEConv(..., 'mean') (model.py:303-308)")
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats = out_feats
        self.feat_drop = nn.Dropout(feat_drop)
        self.norm, self.activation = norm, activation
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_self = nn.Linear(in_feats, out_feats, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph, feat, edge_weight=None):
        feat_src = self.feat_drop(feat)
        feat_dst = feat_src[: graph.num_dst_nodes()]
        lin_before_mp = self._in_src_feats > self._out_feats
        if lin_before_mp:
            h_neigh = weighted_aggregate(graph, self.fc_neigh(feat_src), edge_weight, mean=True)
        else:
            h_neigh = self.fc_neigh(weighted_aggregate(graph, feat_src, edge_weight, mean=True))
        rst = self.fc_self(feat_dst) + h_neigh
        if self.activation is not None:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst


class GraphConv(nn.Module):
    """dglnn.GraphConv(in, out, norm='both', weight=True, bias=True) as built at model.py:397-417.

    [DGL-recalled] forward(graph, feat, edge_weight): feat_src * out_deg^-1/2 (degrees clamped to >= 1); weight first
    iff in > out; aggregate = update_all(u_mul_e('h','_edge_weight'), sum); then weight; * in_deg^-1/2; + bias;
    activation.  Degrees are the block's structural degrees (edge weights do not enter the normalisation)."""

    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None, allow_zero_in_degree=False):
        super().__init__()
        if norm != "both" or not weight:
            raise NotImplementedError("the reference only builds GraphConv(norm='both', weight=True)")
        self._in_feats, self._out_feats, self._activation = in_feats, out_feats, activation
        self._allow_zero_in_degree = allow_zero_in_degree
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats)) if bias else None
        nn.init.xavier_uniform_(self.weight)

    def forward(self, graph, feat, weight=None, edge_weight=None):
        n_dst = graph.num_dst_nodes()
        out_deg = torch.zeros(graph.num_src_nodes(), dtype=torch.int32, device=feat.device)
        out_deg.index_add_(0, graph.src, torch.ones_like(graph.src))
        norm_src = out_deg.clamp(min=1).to(feat.dtype).pow(-0.5)
        feat_src = feat * norm_src[:, None]
        w = self.weight
        if self._in_feats > self._out_feats:
            rst = weighted_aggregate(graph, feat_src @ w, edge_weight, mean=False)
        else:
            rst = weighted_aggregate(graph, feat_src, edge_weight, mean=False) @ w
        in_deg = (graph.indptr[1:n_dst + 1] - graph.indptr[:n_dst]).clamp(min=1).to(feat.dtype).pow(-0.5)
        rst = rst * in_deg[:, None]
        if self.bias is not None:
            rst = rst + self.bias
        if self._activation is not None:
            rst = self._activation(rst)
        return rst
