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
    """``th.reshape(th.norm(h, dim=1, keepdim=True), (-1,))`` of model.py:318-320 -> bf16 [K]  (``torch.ops.bliss.embed_norm``)."""
    from . import ops
    h = h.detach()
    if h.dtype != torch.bfloat16:
        h = h.bfloat16()
    if h.stride(1) != 1:
        h = h.contiguous()
    return ops.embed_norm(h)


def _wait_block(block):
    # A block whose build was left to another stream (train.PipelinedTrainStep: the input layer's block is built beside the
    # first transform of the forward pass): its first consumer waits for the builder's flag.  Set during graph capture only.
    r = getattr(block, "_ready", None)
    if r is not None:
        _lib.check(_lib.lib.bliss_flag_wait(r[0], r[1], _stream()), "bliss_flag_wait")
        block._ready = None


def weighted_aggregate(block, h, edge_weight=None, mean=True, out_fp32=False):
    """out[i] = (1/deg_i if mean) * sum_{e -> i} w_e h[src_e] over a Block -- ``torch.ops.bliss.spmm`` (bliss_gnn_amd/ops.py: a
    torch.library custom op with a fake kernel and the transposed SpMM as its autograd formula); gradient w.r.t. h only
    (the sampler's edge weights carry no grad)."""
    from . import ops
    assert h.is_cuda and h.dtype == torch.bfloat16, "bf16 features on the GPU (load_graph.py:7)"
    _wait_block(block)
    if h.stride(1) != 1:
        h = h.contiguous()
    w = None
    if edge_weight is not None:
        w = edge_weight.reshape(-1)
        if w.dtype != torch.bfloat16:
            w = w.bfloat16()
        w = w.contiguous()
        assert w.numel() == block.num_edges()
    counts = getattr(block, "_counts_dev", None) if block._nnz_ptr else None
    need_t = h.requires_grad and torch.is_grad_enabled()
    t_indptr, t_edge = block.transposed() if need_t else (None, None)
    return ops.spmm(block.indptr, block.src, block.dst, w, h, block.num_dst_nodes(), counts, bool(mean), bool(out_fp32), t_indptr, t_edge)


class _SageEpilogue(torch.autograd.Function):
    """out = dropout_p(relu(a + b)), norm = ||out||_2 per row in one kernel (csrc/spmm.hip: k_sage_epilogue)."""

    @staticmethod
    def forward(ctx, a, b, p, ctr, seed):
        ctx.set_materialize_grads(False)          # (no zero-filled gradients for the outputs nothing differentiates through)
        a, b = a.contiguous(), b.contiguous()
        n, d = a.shape
        out = torch.empty_like(a)
        norm = torch.empty(n, dtype=torch.bfloat16, device=a.device)
        _lib.check(_lib.lib.bliss_sage_epilogue_fwd(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), n, d, float(p), int(seed),
                                                    0 if ctr is None else ctr.data_ptr(), out.data_ptr(), out.stride(0),
                                                    norm.data_ptr(), _stream()), "bliss_sage_epilogue_fwd")
        ctx.save_for_backward(out)
        ctx.p = float(p)
        ctx.mark_non_differentiable(norm)
        return out, norm

    @staticmethod
    def backward(ctx, dout, _dnorm):
        (out,) = ctx.saved_tensors
        if dout is None:
            return None, None, None, None, None
        dout = dout.contiguous()
        if dout.dtype != torch.bfloat16:
            dout = dout.bfloat16()
        din = torch.empty_like(out)
        _lib.check(_lib.lib.bliss_sage_epilogue_bwd(dout.data_ptr(), dout.stride(0), out.data_ptr(), out.stride(0), out.shape[0],
                                                    out.shape[1], ctx.p, din.data_ptr(), din.stride(0), _stream()),
                   "bliss_sage_epilogue_bwd")
        return din, din, None, None, None


def sage_epilogue(self_out, neigh, p, ctr, seed):
    """The tail of a hidden SAGE layer (model.py:321-333) and the row norms the next layer stores (:318-320)."""
    return _SageEpilogue.apply(self_out, neigh, p, ctr, seed)


# ------------------------------------------------------------------------------------------------ loss (csrc/loss.hip)
class _CrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) on bf16 logits, forward and gradient in one launch (train_lightning.py:77-79, :142)."""

    @staticmethod
    def forward(ctx, logits, labels, state):
        x = logits if logits.stride(1) == 1 else logits.contiguous()
        n, c = x.shape
        dx = torch.empty(n, c, dtype=torch.bfloat16, device=x.device)
        rows = torch.empty(n, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib.bliss_cross_entropy(x.data_ptr(), x.stride(0), labels.data_ptr(), n, c, rows.data_ptr(), dx.data_ptr(), dx.stride(0),
                                                loss.data_ptr(), state.data_ptr(), state.data_ptr() + 4, _stream()), "bliss_cross_entropy")
        ctx.save_for_backward(dx)
        return loss.to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g.to(dx.dtype), None, None


def _ce_launch(x, labels, state, x2=None, label_ids=None):
    n, c = x.shape
    dx = torch.empty(n, c, dtype=torch.bfloat16, device=x.device)
    rows = torch.empty(n, dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    if x2 is None and label_ids is None:
        _lib.check(_lib.lib.bliss_cross_entropy(x.data_ptr(), x.stride(0), labels.data_ptr(), n, c, rows.data_ptr(), dx.data_ptr(),
                                                dx.stride(0), loss.data_ptr(), state.data_ptr(), state.data_ptr() + 4, _stream()),
                   "bliss_cross_entropy")
    else:
        _lib.check(_lib.lib.bliss_cross_entropy_sum(x.data_ptr(), x.stride(0), 0 if x2 is None else x2.data_ptr(),
                                                    0 if x2 is None else x2.stride(0), labels.data_ptr(),
                                                    0 if label_ids is None else label_ids.data_ptr(), n, c, rows.data_ptr(), dx.data_ptr(),
                                                    dx.stride(0), loss.data_ptr(), state.data_ptr(), state.data_ptr() + 4, _stream()),
                   "bliss_cross_entropy_sum")
    return loss, dx


class CrossEntropyLoss(nn.Module):
    """``nn.CrossEntropyLoss()`` as the reference builds it (train_lightning.py:77-79): mean over the batch, class-index
    targets.  bf16 logits on the GPU take the one-launch kernel; anything else goes to torch's functional form."""

    def forward(self, logits, target):
        if logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2 and target.dtype == torch.int64 and target.dim() == 1:
            if getattr(self, "_state", None) is None or self._state.device != logits.device:
                self._state = torch.zeros(2, dtype=torch.int32, device=logits.device)       # [0] ticket, [1] error word
            return _CrossEntropy.apply(logits, target.contiguous(), self._state)
        return torch.nn.functional.cross_entropy(logits, target)

    def _eligible(self, logits, target):
        return logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2 and target.dtype == torch.int64 and target.dim() == 1

    def backward_from(self, logits, target):
        """loss.backward() without the loss node: the kernel produces d loss / d logits with the loss, so the train loops call
        ``logits.backward(that)`` directly (saves the ones-fill, the scalar cast and the elementwise product of the generic
        route).  Returns the loss (fp32 scalar, detached)."""
        if not self._eligible(logits, target):
            loss = self.forward(logits, target)
            loss.backward()
            return loss.detach()
        if getattr(self, "_state", None) is None or self._state.device != logits.device:
            self._state = torch.zeros(2, dtype=torch.int32, device=logits.device)
        x = logits.detach()
        loss, dx = _ce_launch(x if x.stride(1) == 1 else x.contiguous(), target.contiguous(), self._state)
        logits.backward(dx)
        return loss

    def backward_from_parts(self, a, b, label_table, label_ids):
        """backward_from for logits = a + b that are never formed (the output layer's fc_self + h_neigh) and labels
        = label_table[label_ids] that are never gathered: one launch, then the same gradient into both addends."""
        ok = (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape == b.shape and a.dim() == 2
              and a.stride(1) == 1 and b.stride(1) == 1 and label_table.dtype == torch.int64 and label_table.dim() == 1
              and label_table.is_contiguous() and label_ids.dtype == torch.int32 and label_ids.is_contiguous()
              and label_ids.numel() == a.shape[0])
        if not ok:
            return self.backward_from(a + b, torch.index_select(label_table, 0, label_ids.long()))
        if getattr(self, "_state", None) is None or self._state.device != a.device:
            self._state = torch.zeros(2, dtype=torch.int32, device=a.device)
        loss, dx = _ce_launch(a.detach(), label_table, self._state, x2=b.detach(), label_ids=label_ids)
        torch.autograd.backward([a, b], [dx, dx])
        return loss


# ------------------------------------------------------------------------------------------------ MFMA tile GEMM (csrc/sage.hip)
TILE_GEMM_MAX_K, TILE_GEMM_MAX_N = 1024, 256


class LazyRows:
    """``table[ids]`` not gathered yet: ``block.srcdata.lazy('features')``.  SAGE.forward hands it to the fused transform,
    whose A-operand load IS the gather (train_lightning.py:138 + model.py:318-329 in one launch)."""

    def __init__(self, table, ids, frame=None, key=None):
        self.table, self.ids, self._frame, self._key = table, ids, frame, key
        self.shape, self.dtype, self.device = (ids.numel(), table.shape[1]), table.dtype, table.device
        self.is_cuda = table.is_cuda

    def materialize(self):
        """The rows as a tensor: through the owning frame when there is one (its gather kernel also leaves the row norms the
        model asks for next, graph._LazyFrame), else a plain index_select."""
        if self._frame is not None:
            return self._frame[self._key]
        return torch.index_select(self.table, 0, self.ids)


def _tg_args(a1, w1, out, m_bound, ids=None, a2=None, w2=None, bias=None, m_dev=0, a_copy=None, in_norm=None, out_norm=None,
             relu=False, p=0.0, seed=0, ctr=None):
    import ctypes as C
    t = _lib.TileGemm()
    t.a1, t.a1_stride, t.ids = a1.data_ptr(), a1.stride(0), 0 if ids is None else ids.data_ptr()
    t.w1, t.w1_stride, t.k1 = w1.data_ptr(), w1.stride(0), w1.shape[1]
    if a2 is not None:
        t.a2, t.a2_stride, t.w2, t.w2_stride, t.k2 = a2.data_ptr(), a2.stride(0), w2.data_ptr(), w2.stride(0), w2.shape[1]
    t.bias = 0 if bias is None else bias.data_ptr()
    t.m_bound, t.m_dev, t.n = int(m_bound), int(m_dev), w1.shape[0]
    t.out, t.out_stride = out.data_ptr(), out.stride(0)
    if a_copy is not None:
        t.a_copy, t.copy_stride = a_copy.data_ptr(), a_copy.stride(0)
    t.in_norm = 0 if in_norm is None else in_norm.data_ptr()
    t.out_norm = 0 if out_norm is None else out_norm.data_ptr()
    t.relu, t.drop_p, t.drop_seed, t.drop_ctr = int(relu), float(p), int(seed) & 0xFFFFFFFF, 0 if ctr is None else ctr.data_ptr()
    return t


def _tile_gemm(first, second=None):
    import ctypes as C
    _lib.check(_lib.lib.bliss_tile_gemm(C.byref(first), None if second is None else C.byref(second), _stream()), "bliss_tile_gemm")


def _bf16c(t):
    t = t if t.dtype == torch.bfloat16 else t.bfloat16()
    return t if t.stride(-1) == 1 else t.contiguous()



# ------------------------------------------------------------------------------------------------ backward products (csrc/sage_bwd.hip)
def _mfma_bwd_on():
    """The hand-written MFMA backward (round 3).  BLISS_SAGE_MFMA_BWD=0 restores the library GEMMs of round 2."""
    import os
    return os.environ.get("BLISS_SAGE_MFMA_BWD", "1") != "0"


def _bwd_ok(*mats):
    return all(m is None or (m.is_cuda and m.dtype == torch.bfloat16 and m.dim() == 2 and m.stride(1) == 1) for m in mats)


def sage_dgrad(a1, w1, m_bound, m_dev=0, a2=None, w2=None, m2_bound=0, m2_dev=0):
    """out[r] = a1[r] @ w1 (+ a2[r] @ w2 for r < m2): the gradient w.r.t. a layer's input rows; w = nn.Linear weights [out, in]."""
    import ctypes as C
    out = torch.empty(m_bound, w1.shape[1], dtype=torch.bfloat16, device=a1.device)
    t = _lib.DGrad()
    t.a1, t.a1_stride, t.w1, t.w1_stride, t.k1 = a1.data_ptr(), a1.stride(0), w1.data_ptr(), w1.stride(0), w1.shape[0]
    if a2 is not None:
        t.a2, t.a2_stride, t.w2, t.w2_stride, t.k2 = a2.data_ptr(), a2.stride(0), w2.data_ptr(), w2.stride(0), w2.shape[0]
        t.m2_bound, t.m2_dev = int(m2_bound), int(m2_dev)
    t.m_bound, t.m_dev, t.n = int(m_bound), int(m_dev), w1.shape[1]
    t.out, t.out_stride = out.data_ptr(), out.stride(0)
    _lib.check(_lib.lib.bliss_sage_dgrad(C.byref(t), _stream()), "bliss_sage_dgrad")
    return out


def sage_wgrad(problems):
    """[(d [R, n_out], x [R, k_in], rows_bound, rows_dev, want_bias)] (one or two) -> [(dW [n_out, k_in], db or None)]: the weight
    (and bias) gradients of Linear layers, dW = d^T x over the block's rows, one launch pair for all of them."""
    import ctypes as C
    n = len(problems)
    arr = (_lib.WGrad * n)()
    outs = []
    for i, (d, x, rb, rdev, want_b) in enumerate(problems):
        dw = torch.empty(d.shape[1], x.shape[1], dtype=torch.bfloat16, device=d.device)
        db = torch.empty(d.shape[1], dtype=torch.bfloat16, device=d.device) if want_b else None
        a = arr[i]
        a.d, a.d_stride, a.n_out = d.data_ptr(), d.stride(0), d.shape[1]
        a.x, a.x_stride, a.k_in = x.data_ptr(), x.stride(0), x.shape[1]
        a.rows_bound, a.rows_dev = int(rb), int(rdev)
        a.dw, a.dw_stride, a.db = dw.data_ptr(), dw.stride(0), 0 if db is None else db.data_ptr()
        outs.append((dw, db))
    need = int(_lib.lib.bliss_sage_wgrad_workspace(arr, n))
    if need < 0:
        raise RuntimeError("bliss_sage_wgrad: invalid problem description")
    # fp32 partial tiles: a scratch tensor of the caching allocator (inside a captured graph: of the graph's pool); the
    # reduce kernel behind the product reads what the product wrote, in stream order, so nothing outlives the call
    ws = torch.empty(max(need, 4), dtype=torch.float32, device=problems[0][0].device)
    _lib.check(_lib.lib.bliss_sage_wgrad(arr, n, ws.data_ptr(), ws.numel(), _stream()), "bliss_sage_wgrad")
    return outs


class _SageLinearPair(torch.autograd.Function):
    """fc_neigh over all source rows and fc_self (+bias) over the destination rows of a W-first SAGEConv layer (in > out,
    [DGL-recalled] SURVEY.md m2/m4) in ONE launch; with ``ids`` the rows are gathered from ``x`` (a node-feature table) on
    the fly.  Returns (Z [K, out], Y [S, out], the input rows [K, in], their bf16 norms [K])."""

    @staticmethod
    def forward(ctx, x, ids, w_neigh, w_self, b_self, n_src, n_dst, src_dev, dst_dev):
        ctx.set_materialize_grads(False)          # (else autograd zero-fills a gradient for `rows`: 13 MB on the input layer)
        x, wn, ws = _bf16c(x), _bf16c(w_neigh), _bf16c(w_self)
        dev, n_out = x.device, wn.shape[0]
        z = torch.empty(n_src, n_out, dtype=torch.bfloat16, device=dev)
        y = torch.empty(n_dst, n_out, dtype=torch.bfloat16, device=dev)
        norm = torch.empty(n_src, dtype=torch.bfloat16, device=dev)
        rows = torch.empty(n_src, x.shape[1], dtype=torch.bfloat16, device=dev) if ids is not None else x
        _tile_gemm(_tg_args(x, wn, z, n_src, ids=ids, m_dev=src_dev, a_copy=rows if ids is not None else None, in_norm=norm),
                   _tg_args(x, ws, y, n_dst, ids=ids, bias=b_self, m_dev=dst_dev))
        ctx.save_for_backward(rows, wn, ws)
        ctx.gathered, ctx.n_dst, ctx.has_bias = ids is not None, n_dst, b_self is not None
        ctx.n_src, ctx.src_dev, ctx.dst_dev = n_src, src_dev, dst_dev
        ctx.mark_non_differentiable(rows, norm) if ids is not None else ctx.mark_non_differentiable(norm)
        return z, y, rows, norm

    @staticmethod
    def backward(ctx, dz, dy, _drows, _dnorm):
        rows, wn, ws = ctx.saved_tensors
        want_dx = not ctx.gathered and ctx.needs_input_grad[0]
        d_wn = d_ws = d_b = dx = None
        if (_mfma_bwd_on() and dz is not None and dy is not None and wn.shape[0] <= 256 and _bwd_ok(rows, wn, ws)
                and dz.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16):
            # csrc/sage_bwd.hip: both weight gradients and the bias gradient in one launch pair, the input gradient
            # (both Linears into one accumulator) in one launch
            dz, dy = _bf16c(dz), _bf16c(dy)
            (d_wn, _), (d_ws, d_b) = sage_wgrad([(dz, rows, ctx.n_src, ctx.src_dev, False), (dy, rows, ctx.n_dst, ctx.dst_dev, ctx.has_bias)])
            if want_dx:
                dx = sage_dgrad(dz, wn, ctx.n_src, ctx.src_dev, a2=dy, w2=ws, m2_bound=ctx.n_dst, m2_dev=ctx.dst_dev)
                if _drows is not None:
                    dx = dx + _drows
            return dx, None, d_wn, d_ws, d_b, None, None, None, None
        if dz is not None:
            dz = _bf16c(dz)
            d_wn = _weight_grad(dz, rows)
            if want_dx:
                dx = dz @ wn
        if dy is not None:
            dy = _bf16c(dy)
            d_ws = _weight_grad(dy, rows[: ctx.n_dst])
            d_b = _bias_grad(dy) if ctx.has_bias else None
            if want_dx:
                if dx is None:
                    dx = torch.zeros_like(rows)
                dx[: ctx.n_dst].addmm_(dy, ws)                                   # (the GEMM accumulates: no separate add)
        if _drows is not None and want_dx:
            dx = _drows if dx is None else dx + _drows
        return dx, None, d_wn, d_ws, d_b, None, None, None, None


class _SageLinearSplit(torch.autograd.Function):
    """_SageLinearPair for a block whose destinations are NOT its leading source rows (a shard's block: the destinations are
    this rank's seeds, anywhere in the global source list): fc_neigh over the source rows and fc_self (+bias) over the
    destination rows -- handed in beside them, or gathered from the source rows through ``dst_ids`` by the launch itself --,
    ONE launch; the input rows' bf16 norms come with it.  Backward on csrc/sage_bwd.hip: both weight gradients and the bias
    gradient in one launch pair, an input gradient per operand."""

    @staticmethod
    def forward(ctx, x_src, x_dst, w_neigh, w_self, b_self, src_dev, dst_dev, dst_ids=None):
        ctx.set_materialize_grads(False)
        xs, wn, ws = _bf16c(x_src), _bf16c(w_neigh), _bf16c(w_self)
        dev, n_out = xs.device, wn.shape[0]
        gathered = x_dst is None
        n_dst = dst_ids.numel() if gathered else x_dst.shape[0]
        xd = torch.empty(n_dst, xs.shape[1], dtype=torch.bfloat16, device=dev) if gathered else _bf16c(x_dst)
        z = torch.empty(xs.shape[0], n_out, dtype=torch.bfloat16, device=dev)
        y = torch.empty(n_dst, n_out, dtype=torch.bfloat16, device=dev)
        norm = torch.empty(xs.shape[0], dtype=torch.bfloat16, device=dev)
        second = (_tg_args(xs, ws, y, n_dst, ids=dst_ids, bias=b_self, m_dev=dst_dev, a_copy=xd) if gathered
                  else _tg_args(xd, ws, y, n_dst, bias=b_self, m_dev=dst_dev))
        _tile_gemm(_tg_args(xs, wn, z, xs.shape[0], m_dev=src_dev, in_norm=norm), second)
        ctx.save_for_backward(xs, xd, wn, ws, dst_ids if gathered else None)
        ctx.has_bias, ctx.src_dev, ctx.dst_dev, ctx.gathered = b_self is not None, src_dev, dst_dev, gathered
        ctx.mark_non_differentiable(norm)
        return z, y, norm

    @staticmethod
    def backward(ctx, dz, dy, _dnorm):
        xs, xd, wn, ws, ids = ctx.saved_tensors
        d_wn = d_ws = d_b = dxs = dxd = None
        probs = []
        if dz is not None:
            dz = _bf16c(dz)
            probs.append((dz, xs, xs.shape[0], ctx.src_dev, False))
        if dy is not None:
            dy = _bf16c(dy)
            probs.append((dy, xd, xd.shape[0], ctx.dst_dev, ctx.has_bias))
        if probs:
            outs = sage_wgrad(probs)
            if dz is not None:
                d_wn = outs[0][0]
            if dy is not None:
                d_ws, d_b = outs[-1]
        if dz is not None and ctx.needs_input_grad[0]:
            dxs = sage_dgrad(dz, wn, xs.shape[0], ctx.src_dev)
        if dy is not None and ctx.needs_input_grad[0 if ctx.gathered else 1]:
            dxd = sage_dgrad(dy, ws, xd.shape[0], ctx.dst_dev)           # (rows at or beyond the true count: zeros)
            if ctx.gathered:                                             # the destination rows came out of x_src: back into its gradient
                dxs = torch.empty_like(xs).fill_(0) if dxs is None else dxs
                dxs.index_add_(0, ids, dxd)                              # (the padding ids repeat row 0 and carry zero rows)
                dxd = None
        return dxs, dxd, d_wn, d_ws, d_b, None, None, None


_ones_rows = {}


def _bias_grad(d):
    """d.sum(0) (the bias gradient of a Linear layer) as a [1, rows] x [rows, out] product with a row of ones: fp32 accumulation,
    one rounding, like aten::sum -- which runs this tall reduction as a memset plus an atomic kernel (15.7 vs 9.8 us for
    5 K x 256, 11.7 vs 6.5 for 2 K x 256; scratch/biasbench.py)."""
    if not (d.is_cuda and d.dim() == 2 and d.shape[0] >= 1024):
        return d.sum(0)
    key = (d.device.index, d.dtype)
    buf = _ones_rows.get(key)
    if buf is None or buf.shape[1] < d.shape[0]:
        buf = _ones_rows[key] = torch.ones(1, max(d.shape[0], 16384), dtype=d.dtype, device=d.device)
    return (buf[:, : d.shape[0]] @ d)[0]


def _weight_grad(d, x, split=4):
    """d.t() @ x for a long reduction (thousands of block rows) and a small result (out x in features): the library runs it on
    out/64 x in/64 = 40 workgroups, a sixth of the chip.  Split the rows into ``split`` batches (fp32 partial products, one
    rounding at the end: the bits of the plain call): 44 -> 32 us for 11 K x 256 x 602, 22.5 -> 19.6 for 5 K rows
    (scratch/dwbench.py, graph replay)."""
    R = d.shape[0]
    if R < 8192 or x.shape[1] < 512 or R % split or not d.is_cuda:       # (in the loop the 5 K-row product is no faster split)
        return d.t() @ x
    p = torch.bmm(d.view(split, R // split, d.shape[1]).transpose(1, 2), x.reshape(split, R // split, x.shape[1]), out_dtype=torch.float32)
    return p.sum(0).to(d.dtype)


class _SageDualLinear(torch.autograd.Function):
    """An aggregate-first SAGEConv layer's tail (in <= out): fc_neigh(h_neigh) + fc_self(h_dst) + bias, ReLU, dropout and the
    row norms of the result (model.py:321-333, :318-320 of the next layer) in ONE launch: both products accumulate in the
    same fp32 registers, one rounding to bf16."""

    @staticmethod
    def forward(ctx, a_neigh, a_self, w_neigh, w_self, bias, relu, p, ctr, seed, n_rows, rows_dev):
        ctx.set_materialize_grads(False)
        a1, a2, w1, w2 = _bf16c(a_neigh), _bf16c(a_self), _bf16c(w_neigh), _bf16c(w_self)
        out = torch.empty(n_rows, w1.shape[0], dtype=torch.bfloat16, device=a1.device)
        norm = torch.empty(n_rows, dtype=torch.bfloat16, device=a1.device)
        _tile_gemm(_tg_args(a1, w1, out, n_rows, a2=a2, w2=w2, bias=bias, m_dev=rows_dev, out_norm=norm, relu=relu, p=p, seed=seed, ctr=ctr))
        ctx.save_for_backward(a1, a2, w1, w2, out)
        ctx.relu, ctx.p, ctx.has_bias, ctx.rows_dev = relu, float(p), bias is not None, rows_dev
        ctx.mark_non_differentiable(norm)
        return out, norm

    @staticmethod
    def backward(ctx, dout, _dnorm):
        a1, a2, w1, w2, out = ctx.saved_tensors
        if dout is None:
            return (None,) * 11
        d = _bf16c(dout)
        if ctx.relu or ctx.p > 0:
            din = torch.empty_like(out)
            _lib.check(_lib.lib.bliss_sage_epilogue_bwd(d.data_ptr(), d.stride(0), out.data_ptr(), out.stride(0), out.shape[0],
                                                        out.shape[1], ctx.p, din.data_ptr(), din.stride(0), _stream()),
                       "bliss_sage_epilogue_bwd")
            d = din
        if _mfma_bwd_on() and w1.shape[0] <= 256 and _bwd_ok(d, a1, a2, w1, w2) and a1.shape[0] == d.shape[0] and a2.shape[0] >= d.shape[0]:
            rd = ctx.rows_dev
            (dw1, _), (dw2, db) = sage_wgrad([(d, a1, d.shape[0], rd, False), (d, a2, d.shape[0], rd, ctx.has_bias)])
            da1 = sage_dgrad(d, w1, d.shape[0], rd) if ctx.needs_input_grad[0] else None
            da2 = None
            if ctx.needs_input_grad[1]:
                da2 = sage_dgrad(d, w2, d.shape[0], rd)
                if a2.shape[0] > d.shape[0]:
                    da2 = torch.cat([da2, torch.zeros(a2.shape[0] - d.shape[0], da2.shape[1], dtype=da2.dtype, device=da2.device)])
            return (da1, da2, dw1, dw2, db, None, None, None, None, None, None)
        return (d @ w1, d @ w2, d.t() @ a1, d.t() @ a2, _bias_grad(d) if ctx.has_bias else None, None, None, None, None, None, None)


class _SageAggDual(torch.autograd.Function):
    """Aggregation + _SageDualLinear as ONE autograd node: the layer input h feeds both the SpMM and (its first rows) fc_self, and
    two nodes made autograd zero-pad the slice's gradient and add the two (fill + copy + add kernels, 5 K x 256 each); here the
    fc_self product accumulates straight into the transposed SpMM's result."""

    @staticmethod
    def forward(ctx, h, indptr, src, dst, w, counts, t_indptr, t_edge, n_dst, w_neigh, w_self, bias, relu, p, ctr, seed, rows_dev):
        from . import ops
        ctx.set_materialize_grads(False)
        h, w1, w2 = _bf16c(h), _bf16c(w_neigh), _bf16c(w_self)
        agg = ops.spmm(indptr, src, dst, w, h, n_dst, counts, True, False, None, None)
        out = torch.empty(n_dst, w1.shape[0], dtype=torch.bfloat16, device=h.device)
        norm = torch.empty(n_dst, dtype=torch.bfloat16, device=h.device)
        _tile_gemm(_tg_args(agg, w1, out, n_dst, a2=h, w2=w2, bias=bias, m_dev=rows_dev, out_norm=norm, relu=relu, p=p, seed=seed, ctr=ctr))
        ctx.save_for_backward(agg, h, w1, w2, out, indptr, src, dst, w, counts, t_indptr, t_edge)
        ctx.relu, ctx.p, ctx.has_bias, ctx.n_dst, ctx.rows_dev = relu, float(p), bias is not None, n_dst, rows_dev
        ctx.mark_non_differentiable(norm)
        return out, norm

    @staticmethod
    def backward(ctx, dout, _dnorm):
        from . import ops
        agg, h, w1, w2, out, indptr, src, dst, w, counts, t_indptr, t_edge = ctx.saved_tensors
        if dout is None:
            return (None,) * 17
        d = _bf16c(dout)
        if ctx.relu or ctx.p > 0:
            din = torch.empty_like(out)
            _lib.check(_lib.lib.bliss_sage_epilogue_bwd(d.data_ptr(), d.stride(0), out.data_ptr(), out.stride(0), out.shape[0],
                                                        out.shape[1], ctx.p, din.data_ptr(), din.stride(0), _stream()),
                       "bliss_sage_epilogue_bwd")
            d = din
        gh = None
        mfma = _mfma_bwd_on() and w1.shape[0] <= 256 and _bwd_ok(d, agg, h, w1, w2)
        if ctx.needs_input_grad[0]:
            if t_indptr is None or t_edge is None:
                raise RuntimeError("the block's by-source index (Block.transposed()) is needed to differentiate w.r.t. h")
            if mfma:
                # A^T (d W_neigh) = (A^T d) W_neigh: the transposed aggregation first, then ONE launch for both Linears' input
                # gradients -- gh = T W_neigh + (d W_self on the destination rows), fp32 accumulation, one rounding
                t = ops.spmm_t(t_indptr, t_edge, src, dst, indptr, w, d, h.shape[0], counts, True)
                src_dev = counts.data_ptr() + 12 if counts is not None else 0
                gh = sage_dgrad(t, w1, h.shape[0], src_dev, a2=d, w2=w2, m2_bound=ctx.n_dst, m2_dev=ctx.rows_dev)
            else:
                gh = ops.spmm_t(t_indptr, t_edge, src, dst, indptr, w, d @ w1, h.shape[0], counts, True)
                gh[: ctx.n_dst].addmm_(d, w2)
        if mfma:
            (dw1, _), (dw2, db) = sage_wgrad([(d, agg, ctx.n_dst, ctx.rows_dev, False), (d, h, ctx.n_dst, ctx.rows_dev, ctx.has_bias)])
            return (gh, None, None, None, None, None, None, None, None, dw1, dw2, db, None, None, None, None, None)
        return (gh, None, None, None, None, None, None, None, None, d.t() @ agg, d.t() @ h[: ctx.n_dst],
                _bias_grad(d) if ctx.has_bias else None, None, None, None, None, None)


def sage_agg_dual(block, h, edge_weight, w_neigh, w_self, bias, relu, p, ctr, seed, rows_dev):
    """mean-aggregate ``h`` over ``block`` and apply fc_neigh(h_neigh) + fc_self(h_dst) + bias (+ ReLU, dropout, row norms)."""
    assert h.is_cuda and h.dtype == torch.bfloat16
    _wait_block(block)
    w = None
    if edge_weight is not None:
        w = edge_weight.reshape(-1)
        w = (w if w.dtype == torch.bfloat16 else w.bfloat16()).contiguous()
    counts = getattr(block, "_counts_dev", None) if block._nnz_ptr else None
    need_t = h.requires_grad and torch.is_grad_enabled()
    t_indptr, t_edge = block.transposed() if need_t else (None, None)
    return _SageAggDual.apply(h, block.indptr, block.src, block.dst, w, counts, t_indptr, t_edge, block.num_dst_nodes(), w_neigh, w_self,
                              bias, relu, p, ctr, seed, rows_dev)


def tile_gemm_ok(in_feats, out_feats):
    return in_feats <= TILE_GEMM_MAX_K and out_feats <= TILE_GEMM_MAX_N


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

    def forward(self, graph, feat, edge_weight=None, parts=False):
        """``parts=True`` returns (fc_self(h_dst), h_neigh) un-added, for a caller that fuses the sum with what follows."""
        if isinstance(feat, tuple):                  # (feat_src, feat_dst) like dglnn.SAGEConv: destinations that are not the
            feat_src, feat_dst = self.feat_drop(feat[0]), self.feat_drop(feat[1])      # leading source rows (shard blocks)
        else:
            feat_src = self.feat_drop(feat)
            feat_dst = feat_src[: graph.num_dst_nodes()]
        lin_before_mp = self._in_src_feats > self._out_feats
        if lin_before_mp:
            h_neigh = weighted_aggregate(graph, self.fc_neigh(feat_src), edge_weight, mean=True)
        else:
            h_neigh = self.fc_neigh(weighted_aggregate(graph, feat_src, edge_weight, mean=True))
        if parts and self.activation is None and self.norm is None:
            return self.fc_self(feat_dst), h_neigh
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
        src, ones = graph.src, None
        if getattr(graph, "_nnz_ptr", 0) and getattr(graph, "_counts_dev", None) is not None:
            # capacity-padded (static-shape) block, round 3: only the first B edges exist (B on the device: LayerCounts::B); the
            # entries behind them are stale and must neither count nor index.  Padding rows have no edges: degree 0 -> clamp 1.
            valid = torch.arange(src.numel(), device=feat.device) < graph._counts_dev[4]
            src, ones = torch.where(valid, src, 0), valid.to(torch.int32)
        # (a fill kernel, not torch.zeros: memset nodes of a captured HIP graph were seen stale on replay)
        out_deg = torch.empty(graph.num_src_nodes(), dtype=torch.int32, device=feat.device).fill_(0)
        out_deg.index_add_(0, src, torch.ones_like(src) if ones is None else ones)
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


# ------------------------------------------------------------------------------------------------ GATv2
def _nnz(block):
    return block._nnz_ptr, int(block.src.numel())


class _GatLogits(torch.autograd.Function):
    """e[e,h] = attn[h,:] . leaky_relu(feat[src_e,h,:] + feat[dst_e,h,:])   (model.py:82-86)."""

    @staticmethod
    def forward(ctx, feat, attn, block, H, D, slope):
        feat, attn = feat.contiguous(), attn.reshape(-1).contiguous()
        nnz_ptr, B = _nnz(block)
        e = torch.empty(B, H, dtype=torch.bfloat16, device=feat.device)
        _lib.check(_lib.lib.bliss_gat_logits(block.src.data_ptr(), block.dst.data_ptr(), nnz_ptr, B, feat.data_ptr(), feat.stride(0),
                                             attn.data_ptr(), H, D, float(slope), e.data_ptr(), _stream()), "bliss_gat_logits")
        ctx.save_for_backward(feat, attn)
        ctx.block, ctx.H, ctx.D, ctx.slope = block, H, D, slope
        return e

    @staticmethod
    def backward(ctx, de):
        feat, attn = ctx.saved_tensors
        block, H, D, slope = ctx.block, ctx.H, ctx.D, ctx.slope
        de = de.contiguous().bfloat16()
        nnz_ptr, B = _nnz(block)
        K, S, HD = feat.shape[0], block.num_dst_nodes(), H * D
        part = torch.empty(max(2 * (-(-B // _lib.lib.bliss_gat_chunk_edges())) * HD, 1), dtype=torch.float32, device=feat.device)
        d_attn = torch.zeros(HD, dtype=torch.float32, device=feat.device)
        d_el = torch.empty(K, HD, dtype=torch.bfloat16, device=feat.device)
        d_er = torch.empty(S, HD, dtype=torch.bfloat16, device=feat.device)
        t_indptr, t_edge = block.transposed()
        for which, rp, n_rows, out, da in ((3, t_indptr, K, d_el, 0), (2, block.indptr, S, d_er, d_attn.data_ptr())):
            _lib.check(_lib.lib.bliss_gat_rows(which, rp.data_ptr(), n_rows, t_edge.data_ptr(), block.src.data_ptr(),
                                               block.dst.data_ptr(), nnz_ptr, B, de.data_ptr(), feat.data_ptr(), feat.stride(0),
                                               attn.data_ptr(), H, D, float(slope), out.data_ptr(), out.stride(0),
                                               part.data_ptr(), da, _stream()), "bliss_gat_rows")
        d_feat = d_el
        d_feat[:S] += d_er                                  # shared weights: the destinations are the first S source rows
        return d_feat, d_attn.to(attn.dtype).view(1, H, D), None, None, None, None


class _EdgeSoftmax(torch.autograd.Function):
    """dglnn.functional.edge_softmax over the in-edges of every destination (model.py:88-90)."""

    @staticmethod
    def forward(ctx, e, block, H):
        e = e.contiguous()
        a = torch.empty_like(e)
        _lib.check(_lib.lib.bliss_gat_edge_softmax(block.indptr.data_ptr(), block.num_dst_nodes(), e.data_ptr(), 0, H, 0,
                                                   a.data_ptr(), _stream()), "bliss_gat_edge_softmax")
        ctx.save_for_backward(a)
        ctx.block, ctx.H = block, H
        return a

    @staticmethod
    def backward(ctx, da):
        (a,) = ctx.saved_tensors
        da = da.contiguous().bfloat16()
        de = torch.empty_like(a)
        _lib.check(_lib.lib.bliss_gat_edge_softmax(ctx.block.indptr.data_ptr(), ctx.block.num_dst_nodes(), da.data_ptr(),
                                                   a.data_ptr(), ctx.H, 1, de.data_ptr(), _stream()), "bliss_gat_edge_softmax")
        return de, None, None


class _GatAggregate(torch.autograd.Function):
    """out[i,h,:] = sum_{e -> i} a[e,h] * feat[src_e,h,:]   (update_all(u_mul_e('el','a'), sum), model.py:98)."""

    @staticmethod
    def forward(ctx, a, feat, block, H, D):
        a, feat = a.contiguous(), feat.contiguous()
        nnz_ptr, B = _nnz(block)
        S, HD = block.num_dst_nodes(), H * D
        out = torch.empty(S, HD, dtype=torch.bfloat16, device=feat.device)
        part = torch.empty(max(2 * (-(-B // _lib.lib.bliss_gat_chunk_edges())) * HD, 1), dtype=torch.float32, device=feat.device)
        _lib.check(_lib.lib.bliss_gat_rows(0, block.indptr.data_ptr(), S, 0, block.src.data_ptr(), block.dst.data_ptr(), nnz_ptr, B,
                                           a.data_ptr(), feat.data_ptr(), feat.stride(0), 0, H, D, 0.0, out.data_ptr(),
                                           out.stride(0), part.data_ptr(), 0, _stream()), "bliss_gat_rows")
        ctx.save_for_backward(a, feat)
        ctx.block, ctx.H, ctx.D = block, H, D
        return out

    @staticmethod
    def backward(ctx, dout):
        a, feat = ctx.saved_tensors
        block, H, D = ctx.block, ctx.H, ctx.D
        dout = dout.contiguous().bfloat16()
        nnz_ptr, B = _nnz(block)
        K, HD = feat.shape[0], H * D
        da = torch.empty(B, H, dtype=torch.bfloat16, device=feat.device)
        _lib.check(_lib.lib.bliss_gat_edge_dot(block.src.data_ptr(), block.dst.data_ptr(), nnz_ptr, B, feat.data_ptr(), feat.stride(0),
                                               dout.data_ptr(), dout.stride(0), H, D, da.data_ptr(), _stream()), "bliss_gat_edge_dot")
        t_indptr, t_edge = block.transposed()
        d_feat = torch.empty(K, HD, dtype=torch.bfloat16, device=feat.device)
        part = torch.empty(max(2 * (-(-B // _lib.lib.bliss_gat_chunk_edges())) * HD, 1), dtype=torch.float32, device=feat.device)
        _lib.check(_lib.lib.bliss_gat_rows(1, t_indptr.data_ptr(), K, t_edge.data_ptr(), block.src.data_ptr(), block.dst.data_ptr(),
                                           nnz_ptr, B, a.data_ptr(), dout.data_ptr(), dout.stride(0), 0, H, D, 0.0,
                                           d_feat.data_ptr(), d_feat.stride(0), part.data_ptr(), 0, _stream()), "bliss_gat_rows")
        return da, d_feat, None, None, None


def _gat_segments(block, S, B, state):
    """wg_row int32 [cap_wg, 4] (one descriptor per virtual workgroup), n_wg int32 [1] for ``block`` (bliss_gat_segments) and the per-row words the sharing workgroups
    meet on (zero-initialised once per layer and size; the kernels return them to zero)."""
    import ctypes as C
    dev = block.indptr.device
    cap_wg = S + B // int(_lib.lib.bliss_gat_segment_edges()) + 1
    if state["row_ws"] is None or state["row_ws"].numel() < S * 32:
        state["row_ws"] = torch.zeros(max(S, 1) * 32, dtype=torch.int32, device=dev)
    wg_row = torch.empty(cap_wg, 4, dtype=torch.int32, device=dev)
    n_wg = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib.bliss_gat_segments(block.indptr.data_ptr(), S, cap_wg, wg_row.data_ptr(), n_wg.data_ptr(), state["err"].data_ptr(),
                                           _stream()), "bliss_gat_segments")
    return dict(wg_row=wg_row, n_wg=n_wg, cap_wg=cap_wg)


def _gat_fused_on(H, D):
    import os
    return os.environ.get("BLISS_GAT_FUSED", "1") != "0" and bool(_lib.lib.bliss_gat_fused_supported(int(H), int(D)))


class _GatFusedMP(torch.autograd.Function):
    """model.py:82-99 in ONE launch per direction and destination row (csrc/gat_fused.hip): (rst [S, H*D], e [B, H]) from
    feat = fc_src(h) and attn, attention dropout inside (model.py:88; counter-hash stream like the SAGE epilogue's -- the same
    Bernoulli(1-p)/scale law as nn.Dropout, not torch's Philox stream).  Backward: by destination in one launch pair (d a,
    softmax backward, d er, d attn), by source in one merge-style pass (logits + aggregation backward together)."""

    @staticmethod
    def forward(ctx, feat, attn, block, H, D, slope, p_drop, state):
        import ctypes as C
        ctx.set_materialize_grads(False)
        feat, attn_f = feat.contiguous(), attn.reshape(-1).contiguous()
        nnz_ptr, B = _nnz(block)
        S, HD, dev = block.num_dst_nodes(), H * D, feat.device
        e = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
        a = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
        ad = torch.empty(B, H, dtype=torch.bfloat16, device=dev) if p_drop > 0 else a
        rst = torch.empty(S, HD, dtype=torch.bfloat16, device=dev)
        t = _lib.GatFused()
        t.indptr, t.src, t.n_dst = block.indptr.data_ptr(), block.src.data_ptr(), S
        t.n_dst_dev = block._counts_dev.data_ptr() if (block._nnz_ptr and getattr(block, "_counts_dev", None) is not None) else 0
        t.feat, t.feat_stride, t.attn, t.heads, t.head_dim, t.negative_slope = feat.data_ptr(), feat.stride(0), attn_f.data_ptr(), H, D, float(slope)
        t.e, t.a, t.a_drop, t.rst, t.rst_stride = e.data_ptr(), a.data_ptr(), ad.data_ptr(), rst.data_ptr(), rst.stride(0)
        if p_drop > 0:
            t.drop_p, t.drop_seed, t.drop_ctr = float(p_drop), int(state["seed"]) & 0xFFFFFFFF, state["ctr"].data_ptr()
        # hub destinations are shared by several workgroups: virtual workgroup -> row map of THIS block (one small launch)
        seg = _gat_segments(block, S, B, state)
        seg_part = torch.empty(seg["cap_wg"] * (HD + 8), dtype=torch.float32, device=dev)
        t.wg_row, t.n_wg_dev, t.cap_wg = seg["wg_row"].data_ptr(), seg["n_wg"].data_ptr(), seg["cap_wg"]
        t.row_ws, t.seg_part, t.err = state["row_ws"].data_ptr(), seg_part.data_ptr(), state["err"].data_ptr()
        _lib.check(_lib.lib.bliss_gat_fused_fwd(C.byref(t), _stream()), "bliss_gat_fused_fwd")
        ctx.save_for_backward(feat, attn_f, a, ad, seg["wg_row"], seg["n_wg"])
        ctx.cap_wg = seg["cap_wg"]
        ctx.block, ctx.H, ctx.D, ctx.slope, ctx.p, ctx.state, ctx.n_dst_dev = block, H, D, float(slope), float(p_drop), state, t.n_dst_dev
        ctx.attn_shape, ctx.attn_dtype = attn.shape, attn.dtype
        return rst, e

    @staticmethod
    def backward(ctx, d_rst, _d_e):
        import ctypes as C
        feat, attn_f, a, ad, wg_row, n_wg = ctx.saved_tensors
        if _d_e is not None:
            raise NotImplementedError("the returned logits are the bandit's a_ij (model.py:224-227): nothing differentiates through them")
        if d_rst is None:
            return (None,) * 8
        block, H, D = ctx.block, ctx.H, ctx.D
        nnz_ptr, B = _nnz(block)
        K, S, HD, dev = feat.shape[0], block.num_dst_nodes(), H * D, feat.device
        g = d_rst.reshape(S, HD)
        g = (g if g.dtype == torch.bfloat16 else g.bfloat16()).contiguous()
        de = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
        d_er = torch.empty(S, HD, dtype=torch.bfloat16, device=dev)
        cap_wg = ctx.cap_wg
        dpart = torch.empty(cap_wg, HD, dtype=torch.float32, device=dev)
        bsum = torch.empty(-(-cap_wg // 32), HD, dtype=torch.float32, device=dev)
        seg_part = torch.empty(cap_wg * (HD + 8), dtype=torch.float32, device=dev)
        d_attn = torch.empty(HD, dtype=torch.float32, device=dev)
        t = _lib.GatFused()
        t.indptr, t.src, t.n_dst, t.n_dst_dev = block.indptr.data_ptr(), block.src.data_ptr(), S, ctx.n_dst_dev
        t.feat, t.feat_stride, t.attn, t.heads, t.head_dim, t.negative_slope = feat.data_ptr(), feat.stride(0), attn_f.data_ptr(), H, D, ctx.slope
        t.a, t.a_drop = a.data_ptr(), ad.data_ptr()
        t.drop_p = ctx.p
        t.g, t.g_stride, t.de, t.d_er, t.d_er_stride, t.dattn_part = g.data_ptr(), g.stride(0), de.data_ptr(), d_er.data_ptr(), d_er.stride(0), dpart.data_ptr()
        t.wg_row, t.n_wg_dev, t.cap_wg = wg_row.data_ptr(), n_wg.data_ptr(), cap_wg
        t.row_ws, t.seg_part, t.err = ctx.state["row_ws"].data_ptr(), seg_part.data_ptr(), ctx.state["err"].data_ptr()
        _lib.check(_lib.lib.bliss_gat_fused_bwd_dst(C.byref(t), bsum.data_ptr(), d_attn.data_ptr(), ctx.state["ticket"].data_ptr(), _stream()),
                   "bliss_gat_fused_bwd_dst")
        t_indptr, t_edge = block.transposed()
        d_feat = torch.empty(K, HD, dtype=torch.bfloat16, device=dev)
        part = torch.empty(max(2 * (-(-B // _lib.lib.bliss_gat_chunk_edges())) * HD, 4), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.bliss_gat_rows_src_fused(t_indptr.data_ptr(), K, t_edge.data_ptr(), block.src.data_ptr(), block.dst.data_ptr(), nnz_ptr, B,
                                                     de.data_ptr(), ad.data_ptr(), feat.data_ptr(), feat.stride(0), g.data_ptr(), g.stride(0),
                                                     attn_f.data_ptr(), H, D, ctx.slope, d_feat.data_ptr(), d_feat.stride(0), part.data_ptr(),
                                                     _stream()), "bliss_gat_rows_src_fused")
        d_feat[:S] += d_er                                  # shared weights: the destinations are the first S source rows
        return d_feat, d_attn.to(ctx.attn_dtype).view(ctx.attn_shape), None, None, None, None, None, None


def edge_softmax(graph, logits):
    """``dglnn.functional.edge_softmax(graph, e)`` (model.py:88-90): softmax over the in-edges of every destination,
    per head; ``logits`` [B, H, 1] (or [B, H])."""
    shape = logits.shape
    H = shape[1] if logits.dim() > 1 else 1
    return _EdgeSoftmax.apply(logits.reshape(shape[0], H), graph, H).view(shape)


def gat_forward_f32(graph, feat, attn, H, D, slope):
    """The message-passing part of custom_GATv2Conv.forward (model.py:82-99) WITHOUT the intermediate bf16 roundings of the
    default kernels and with float results: (e [B, H], a [B, H], out [S, H*D]) from the bf16 operands ``feat`` = fc_src(h)
    [K, H*D] and ``attn``.  Forward only -- the check of the north star's 1e-4 bound against fp32 math (tests/); the model
    path rounds where the reference's bf16 tensor ops round."""
    feat, attn = feat.detach().contiguous(), attn.detach().reshape(-1).contiguous()
    assert feat.dtype == torch.bfloat16 and attn.dtype == torch.bfloat16 and feat.is_cuda
    nnz_ptr, B = _nnz(graph)
    S, HD, dev = graph.num_dst_nodes(), H * D, feat.device
    e = torch.empty(B, H, dtype=torch.float32, device=dev)
    a = torch.empty(B, H, dtype=torch.float32, device=dev)
    out = torch.empty(S, HD, dtype=torch.float32, device=dev)
    part = torch.empty(max(2 * (-(-B // _lib.lib.bliss_gat_chunk_edges())) * HD, 1), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib.bliss_gat_logits_f32(graph.src.data_ptr(), graph.dst.data_ptr(), nnz_ptr, B, feat.data_ptr(), feat.stride(0),
                                             attn.data_ptr(), H, D, float(slope), e.data_ptr(), _stream()), "bliss_gat_logits_f32")
    _lib.check(_lib.lib.bliss_gat_edge_softmax(graph.indptr.data_ptr(), S, e.data_ptr(), 0, H, 2, a.data_ptr(), _stream()),
               "bliss_gat_edge_softmax")
    _lib.check(_lib.lib.bliss_gat_rows(4, graph.indptr.data_ptr(), S, 0, graph.src.data_ptr(), graph.dst.data_ptr(), nnz_ptr, B,
                                       a.data_ptr(), feat.data_ptr(), feat.stride(0), 0, H, D, 0.0, out.data_ptr(), out.stride(0),
                                       part.data_ptr(), 0, _stream()), "bliss_gat_rows")
    return e, a, out


class GATv2Conv(nn.Module):
    """The reference's ``custom_GATv2Conv`` (model.py:13-112): dglnn.GATv2Conv with the forward that returns the
    PRE-softmax logits as "attention" (:108-110) and ignores ``edge_weight`` (:91-96 are commented out).

    [DGL-recalled] parameters: fc_src Linear(in, H*D, bias), shared with fc_dst when share_weights; attn (1,H,D);
    res_fc Linear(in, H*D, bias) when residual and in != H*D, identity when equal; xavier_normal(gain('relu'))."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0, negative_slope=0.2, residual=False,
                 activation=None, allow_zero_in_degree=False, bias=True, share_weights=False):
        super().__init__()
        if not share_weights:
            raise NotImplementedError("the reference always builds share_weights=True (model.py:152,170,186,202)")
        self._num_heads, self._out_feats, self._in = num_heads, out_feats, in_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self.fc_src = nn.Linear(in_feats, out_feats * num_heads, bias=bias)
        self.fc_dst = self.fc_src
        self.attn = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop, self.attn_drop = nn.Dropout(feat_drop), nn.Dropout(attn_drop)
        self.negative_slope = negative_slope
        if residual:
            self.res_fc = nn.Linear(in_feats, num_heads * out_feats, bias=bias) if in_feats != out_feats * num_heads else nn.Identity()
        else:
            self.res_fc = None
        self.activation = activation
        self.share_weights = share_weights
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc_src.weight, gain=gain)
        nn.init.xavier_normal_(self.attn, gain=gain)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)
        if bias:
            nn.init.constant_(self.fc_src.bias, 0)

    def _fused_state(self, device):
        """Device state of the fused kernels: the attention-dropout launch counter (uint64[2]: counter, ticket), its seed, and
        the ticket of the d attn reduction.  The seed is a function of torch's CUDA seed and of the layer's ordinal in its model
        (``_drop_salt``, set by model.GATv2): the same seeds give the same masks run after run (round 3: it used to mix in
        ``id(self)``, so two runs of one script trained differently)."""
        st = getattr(self, "_fstate", None)
        if st is None or st["ctr"].device != device:
            st = self._fstate = dict(ctr=torch.zeros(2, dtype=torch.int64, device=device), ticket=torch.zeros(1, dtype=torch.int32, device=device),
                                     seed=(torch.cuda.initial_seed() ^ (0x9E3779B1 * (int(getattr(self, "_drop_salt", 0)) + 1))) & 0xFFFFFFFF,
                                     err=torch.zeros(1, dtype=torch.int32, device=device), row_ws=None)
        return st

    def check_errors(self):
        """A row shared by several workgroups whose partners failed to meet within the spin bound (never seen; the grid drains
        and the word says so)."""
        st = getattr(self, "_fstate", None)
        if st is not None and int(st["err"].item()):
            raise RuntimeError("GATv2 fused kernels: error 0x%x (%s)" % (int(st["err"].item()), _lib.err_string(int(st["err"].item()))))

    def forward(self, graph, feat, edge_weight=None, get_attention=False):
        H, D, S = self._num_heads, self._out_feats, graph.num_dst_nodes()
        if not self._allow_zero_in_degree and bool((graph.in_degrees() == 0).any()):
            raise RuntimeError("There are 0-in-degree nodes in the graph (model.py:49-61); build the layer with "
                               "allow_zero_in_degree=True or add self loops")
        h_src = self.feat_drop(feat)
        feat_src = self.fc_src(h_src)                                            # :66-72, [K, H*D]; feat_dst = feat_src[:S]
        if feat_src.is_cuda and feat_src.dtype == torch.bfloat16 and _gat_fused_on(H, D):
            p_drop = float(self.attn_drop.p) if self.training else 0.0
            rst, e = _GatFusedMP.apply(feat_src, self.attn, graph, H, D, self.negative_slope, p_drop, self._fused_state(feat_src.device))
            rst = rst.view(S, H, D)                                              # :82-99 in one launch
        else:
            e = _GatLogits.apply(feat_src, self.attn, graph, H, D, self.negative_slope)           # :82-86
            a = self.attn_drop(_EdgeSoftmax.apply(e, graph, H))                  # :88-90
            rst = _GatAggregate.apply(a, feat_src, graph, H, D).view(S, H, D)    # :98-99
        if self.res_fc is not None:
            rst = rst + self.res_fc(h_src[:S]).view(S, -1, D)                    # :101-103
        if self.activation:
            rst = self.activation(rst)
        return (rst, e.view(-1, H, 1)) if get_attention else rst                 # :108-112
