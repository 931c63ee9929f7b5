"""The message-passing kernels as ``torch.library`` custom ops (last row): ``torch.ops.bliss.spmm``
(the weighted g-SpMM of dglnn.SAGEConv / GraphConv, [DGL-recalled] SURVEY m3, with its transposed backward m6 registered as
the op's autograd formula) and ``torch.ops.bliss.embed_norm`` (model.py:318-320).  Device tensors in, device tensors out,
no host sync; fake (meta) implementations make them traceable (FakeTensor / torch.compile shape propagation); the real
implementations are the C ABI calls of include/bliss_gnn.h -- there is no other backend.

``counts`` (optional int32[10] = a bliss_layer_counts_t on the device) marks capacity-padded blocks: the true edge count is
read on the device, padded edges are inert (bliss_gnn_amd._engine)."""
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._engine import _stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _nnz_ptr(counts):
    return 0 if counts is None else counts.data_ptr() + 16          # bliss_layer_counts_t::B


@torch.library.custom_op("bliss::spmm", mutates_args=())
def spmm(indptr: Tensor, src: Tensor, dst: Tensor, w: Optional[Tensor], h: Tensor, n_dst: int, counts: Optional[Tensor], mean: bool,
         out_fp32: bool, t_indptr: Optional[Tensor], t_edge: Optional[Tensor]) -> Tensor:
    """out[i] = (1/deg_i if mean) * sum_{e -> i} w_e h[src_e]  -- CSR by destination (indptr, src, dst per edge)."""
    assert h.is_cuda and h.dtype == torch.bfloat16 and h.stride(1) == 1, "bf16 features on the GPU (load_graph.py:7)"
    B, D = int(src.numel()), h.shape[1]
    out = torch.empty(n_dst, D, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=h.device)
    ec = _lib.lib.bliss_spmm_chunk_edges(B)
    part = torch.empty(2 * ((B + ec - 1) // ec) * D, dtype=torch.float32, device=h.device) if B > 0 else None
    _lib.check(_lib.lib.bliss_spmm_fwd(indptr.data_ptr(), src.data_ptr(), dst.data_ptr(), _p(w), h.data_ptr(), h.stride(0), n_dst,
                                       _nnz_ptr(counts), B, D, int(mean), out.data_ptr(), out.stride(0), int(out_fp32), _p(part),
                                       _stream()), "bliss_spmm_fwd")
    return out


@spmm.register_fake
def _(indptr, src, dst, w, h, n_dst, counts, mean, out_fp32, t_indptr, t_edge):
    return h.new_empty((n_dst, h.shape[1]), dtype=torch.float32 if out_fp32 else torch.bfloat16)


@torch.library.custom_op("bliss::spmm_t", mutates_args=())
def spmm_t(t_indptr: Tensor, t_edge: Tensor, src: Tensor, dst: Tensor, indptr: Tensor, w: Optional[Tensor], gout: Tensor, n_src: int,
           counts: Optional[Tensor], mean: bool) -> Tensor:
    """The transposed product: gh[j] = sum_{e : src_e = j} (w_e / deg_{dst_e} if mean) gout[dst_e]  (by-source index t_*)."""
    B, D = int(src.numel()), gout.shape[1]
    gh = torch.empty(n_src, D, dtype=torch.bfloat16, device=gout.device)
    ec = _lib.lib.bliss_spmm_chunk_edges(B)
    part = torch.empty(2 * ((B + ec - 1) // ec) * D, dtype=torch.float32, device=gout.device) if B > 0 else None
    _lib.check(_lib.lib.bliss_spmm_bwd(t_indptr.data_ptr(), t_edge.data_ptr(), src.data_ptr(), dst.data_ptr(), indptr.data_ptr(), _p(w),
                                       gout.data_ptr(), gout.stride(0), n_src, _nnz_ptr(counts), B, D, int(mean), gh.data_ptr(),
                                       gh.stride(0), 0, _p(part), _stream()), "bliss_spmm_bwd")
    return gh


@spmm_t.register_fake
def _(t_indptr, t_edge, src, dst, indptr, w, gout, n_src, counts, mean):
    return gout.new_empty((n_src, gout.shape[1]), dtype=torch.bfloat16)


def _spmm_setup(ctx, inputs, output):
    indptr, src, dst, w, h, n_dst, counts, mean, out_fp32, t_indptr, t_edge = inputs
    ctx.save_for_backward(indptr, src, dst, w, counts, t_indptr, t_edge)
    ctx.n_src, ctx.mean = h.shape[0], mean


def _spmm_backward(ctx, gout):
    indptr, src, dst, w, counts, t_indptr, t_edge = ctx.saved_tensors
    if t_indptr is None or t_edge is None:
        raise RuntimeError("bliss::spmm needs the block's by-source index (Block.transposed()) to differentiate w.r.t. h")
    g = gout.contiguous()
    if g.dtype != torch.bfloat16:
        g = g.bfloat16()
    gh = spmm_t(t_indptr, t_edge, src, dst, indptr, w, g, ctx.n_src, counts, ctx.mean)
    return (None, None, None, None, gh, None, None, None, None, None, None)       # the sampler's edge weights carry no grad (m6)


spmm.register_autograd(_spmm_backward, setup_context=_spmm_setup)


@torch.library.custom_op("bliss::embed_norm", mutates_args=())
def embed_norm(h: Tensor) -> Tensor:
    """bf16 [K]: ||h_j||_2 per row (model.py:318-320)."""
    assert h.is_cuda and h.dtype == torch.bfloat16 and h.stride(1) == 1
    out = torch.empty(h.shape[0], dtype=torch.bfloat16, device=h.device)
    _lib.check(_lib.lib.bliss_embed_norm(h.data_ptr(), h.shape[0], h.shape[1], h.stride(0), out.data_ptr(), _stream()), "bliss_embed_norm")
    return out


@embed_norm.register_fake
def _(h):
    return h.new_empty((h.shape[0],), dtype=torch.bfloat16)
