"""Models over bliss Blocks with the reference's constructor and forward signatures (model.py).

``SAGE(in_feats, n_hidden, n_classes, n_layers, activation, dropout).forward(blocks, x)`` stores the
source-row norms the bandit reward needs on ``block.srcdata['embed_norm']`` (model.py:318-320) and
runs the SAGEConv layers with the sampler's edge weights (model.py:321-329).
"""
import torch
import torch.nn as nn

from .nn import SAGEConv, embed_norm, sage_epilogue


def _gathered_norm(blocks, x):
    """The row norms of ``x`` if it is blocks[0]'s feature slice as gathered by the fused gather + norm kernel
    (graph._LazyFrame), else None (the caller then runs embed_norm: same bits either way)."""
    fn = getattr(blocks[0].srcdata, "row_norm_of", None) if len(blocks) else None
    return fn(x) if fn is not None else None


class SAGE(nn.Module):
    """model.py:292-383."""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout):
        super().__init__()
        self.n_layers, self.n_hidden, self.n_classes = n_layers, n_hidden, n_classes
        self.layers = nn.ModuleList()
        if n_layers > 1:
            self.layers.append(SAGEConv(in_feats, n_hidden, "mean"))
            for _ in range(1, n_layers - 1):
                self.layers.append(SAGEConv(n_hidden, n_hidden, "mean"))
            self.layers.append(SAGEConv(n_hidden, n_classes, "mean"))
        else:
            self.layers.append(SAGEConv(in_feats, n_classes, "mean"))
        self.dropout = nn.Dropout(dropout)
        self.activation = activation

    def _fusable(self, h):
        act = self.activation
        relu = act in (torch.relu, torch.nn.functional.relu) or isinstance(act, nn.ReLU)
        return relu and h.is_cuda and h.dtype == torch.bfloat16 and isinstance(self.dropout, nn.Dropout)

    def _dropout_state(self, l, device):
        """Launch counter of the fused epilogue's dropout stream for layer l (device uint64[2]) and its seed."""
        if not hasattr(self, "_drop_ctr"):
            self._drop_ctr = {}
        key = (l, str(device))
        if key not in self._drop_ctr:
            self._drop_ctr[key] = torch.zeros(2 + 64, dtype=torch.int64, device=device)      # launch counter, ticket, 64 sub-tickets
        return self._drop_ctr[key], (torch.cuda.initial_seed() ^ (0x9E3779B1 * (l + 1))) & 0xFFFFFFFF

    accepts_lazy_rows = True        # forward() takes blocks[0].srcdata.lazy('features'): the gather becomes an operand load

    def _mfma_ok(self, x):
        """The fused matrix-core path (csrc/sage.hip) covers the reference's configuration: bf16 on the GPU, ReLU, every
        layer within the tile kernel's limits (in <= 1024, out <= 256)."""
        import os
        from .nn import tile_gemm_ok
        # BLISS_SAGE_MFMA=0 falls back to library GEMMs + separate gather / epilogue kernels (the round-1 path).  Device time per
        # launch on the Reddit-like step (scratch/tgbench.py, graph replay): 602 -> 256 pair with the gather 32-37 us (gather +
        # two tuned library GEMMs: 46), 256 + 256 dual with epilogue 20 (23), 256 -> 41 pair 13 (18), and five launches fewer
        if os.environ.get("BLISS_SAGE_MFMA", "1") == "0":
            return False
        act = self.activation
        relu = act in (torch.relu, torch.nn.functional.relu) or isinstance(act, nn.ReLU)
        dims = all(tile_gemm_ok(l._in_src_feats, l._out_feats) and l.fc_self.bias is not None and l.norm is None and l.activation is None
                   and l.feat_drop.p == 0 for l in self.layers)
        return relu and dims and x.is_cuda and x.dtype == torch.bfloat16 and isinstance(self.dropout, nn.Dropout)

    def _layers_mfma(self, blocks, h, norm, lo, hi, parts=False):
        """model.py:312-333, layers lo..hi-1, with the Linear layers on hand-written MFMA tiles: per W-first layer ONE launch
        for fc_neigh + fc_self (+ the feature gather and the input norms), per aggregate-first layer ONE launch for both
        Linears, the bias, ReLU, dropout and the next layer's norms.  The aggregation stays the merge-style SpMM of
        csrc/spmm.hip."""
        from .nn import LazyRows, _SageLinearPair, sage_agg_dual, weighted_aggregate
        n_layers = len(self.layers)
        for l in range(lo, hi):
            layer, block = self.layers[l], blocks[l]
            last = l == n_layers - 1
            S_b, K_b = block.num_dst_nodes(), block.num_src_nodes()
            cd = block._counts_dev.data_ptr() if block._nnz_ptr else 0           # capacity-padded block: true S, K on the device
            src_dev, dst_dev = (cd + 12, cd) if cd else (0, 0)
            ew = block.edata["edge_weights"] if "edge_weights" in block.edata else None
            p = self.dropout.p if (self.training and not last) else 0.0
            ctr, seed = self._dropout_state(l, h.device) if p > 0 else (None, 0)
            fc_n, fc_s = layer.fc_neigh, layer.fc_self
            if layer._in_src_feats > layer._out_feats:                           # fc_neigh before the aggregation
                table, ids = (h.table, h.ids) if isinstance(h, LazyRows) else (h, None)
                z, y, _rows, in_norm = _SageLinearPair.apply(table, ids, fc_n.weight, fc_s.weight, fc_s.bias, K_b, S_b, src_dev, dst_dev)
                block.srcdata["embed_norm"] = in_norm if norm is None else norm  # model.py:318-320 (same bits either way)
                agg = weighted_aggregate(block, z, ew, mean=True)
                if last:
                    h, norm = ((y, agg) if parts else y + agg), None          # (parts: the loss kernel adds them itself)
                else:
                    h, norm = sage_epilogue(y, agg, p, ctr, seed)
            else:
                if isinstance(h, LazyRows):
                    h = h.materialize()
                block.srcdata["embed_norm"] = embed_norm(h) if norm is None else norm
                h, norm = sage_agg_dual(block, h, ew, fc_n.weight, fc_s.weight, fc_s.bias, not last, p, ctr, seed, dst_dev)
                if last:
                    norm = None
        return h, norm

    def _layers_plain(self, blocks, h, norm, lo, hi):
        """model.py:312-333, layers lo..hi-1, on library GEMMs (+ the fused epilogue where it applies)."""
        for l in range(lo, hi):
            layer, block = self.layers[l], blocks[l]
            block.srcdata["embed_norm"] = embed_norm(h) if norm is None else norm          # model.py:318-320
            norm = None
            ew = block.edata["edge_weights"] if "edge_weights" in block.edata else None
            if l < len(self.layers) - 1 and self._fusable(h):
                # rst = fc_self + h_neigh, activation, dropout (:321-333) and the next layer's row norms in ONE kernel
                parts = layer(block, h, edge_weight=ew, parts=True)
                if isinstance(parts, tuple):
                    p = self.dropout.p if self.training else 0.0
                    ctr, seed = self._dropout_state(l, h.device) if p > 0 else (None, 0)
                    h, norm = sage_epilogue(parts[0], parts[1], p, ctr, seed)
                    continue
                h = parts
            else:
                h = layer(block, h, edge_weight=ew)
            if l < len(self.layers) - 1:
                h = self.activation(h)
                h = self.dropout(h)
        return h, norm

    def _layers(self, blocks, x, norm, lo, hi, mfma):
        from .nn import LazyRows
        if mfma:
            return self._layers_mfma(blocks, x, norm, lo, hi)
        if isinstance(x, LazyRows):
            x = x.materialize()
        if lo == 0 and norm is None:
            norm = _gathered_norm(blocks, x)
        return self._layers_plain(blocks, x, norm, lo, hi)

    def forward(self, blocks, x):
        return self._layers(blocks, x, None, 0, len(self.layers), self._mfma_ok(x))[0]

    # The bandit update (sampler.exp3) reads every block's INPUT row norms (model.py:318-320) and nothing the output layer
    # computes, so a training loop may run forward_hidden -> exp3 -> next batch's sampler and leave forward_last + loss +
    # backward to a second stream (train.PipelinedTrainStep).  forward(blocks, x) == forward_last(blocks, forward_hidden(...)).
    def forward_hidden(self, blocks, x):
        """Layers 0..L-2 and the output layer's input norms: every block's srcdata['embed_norm'] is set on return."""
        n = len(self.layers)
        if n < 2:
            raise ValueError("forward_hidden needs at least one hidden layer")
        mfma = self._mfma_ok(x)
        h, norm = self._layers(blocks, x, None, 0, n - 1, mfma)
        if norm is None:
            norm = embed_norm(h)
        blocks[n - 1].srcdata["embed_norm"] = norm
        return h, norm, mfma

    def forward_last(self, blocks, hidden):
        """The output layer on forward_hidden's result."""
        h, norm, mfma = hidden
        n = len(self.layers)
        return self._layers(blocks, h, norm, n - 1, n, mfma)[0]

    def forward_last_parts(self, blocks, hidden):
        """forward_last as the two addends of ``rst = fc_self + h_neigh`` (model.py:321-329) when the output layer runs
        fc_neigh before the aggregation on the fused path (the loss kernel then adds them itself); else None."""
        h, norm, mfma = hidden
        n = len(self.layers)
        layer = self.layers[n - 1]
        if not (mfma and layer._in_src_feats > layer._out_feats):
            return None
        return self._layers_mfma(blocks, h, norm, n - 1, n, parts=True)[0]

    @torch.no_grad()
    def inference(self, g, device=None, batch_size=128, use_uva=False, num_workers=0, node_chunk=16384):
        """model.py:335-383: layer-wise full-neighbour inference over ALL nodes (no sampling, plain mean, no edge weights).

        The reference walks the nodes 128 at a time through a DataLoader; every output row depends only on its own
        in-neighbours, so the rows are computed here in chunks of ``node_chunk`` contiguous nodes straight from the CSC
        (``batch_size``, ``use_uva`` and ``num_workers`` are accepted for signature compatibility).  Returns y [V, classes]
        and leaves it in ``g.ndata['h']`` like the reference (:382)."""
        was_training = self.training
        self.eval()                                                          # :364
        V = g.num_nodes()
        deg = g.indptr[1:] - g.indptr[:-1]
        dst_all = torch.repeat_interleave(torch.arange(V, device=g.device, dtype=torch.int32), deg)
        h = g.ndata["features"]                                              # :346
        for l, layer in enumerate(self.layers):                              # :366
            before = layer._in_src_feats > layer._out_feats                  # SAGEConv: fc_neigh before aggregation iff in > out
            src_feat = layer.fc_neigh(h) if before else h
            y = torch.empty(V, layer._out_feats, dtype=h.dtype, device=h.device)
            for b0 in range(0, V, node_chunk):
                b1 = min(V, b0 + node_chunk)
                neigh = _full_neighbor_mean(g, src_feat, b0, b1, dst_all)
                if not before:
                    neigh = layer.fc_neigh(neigh)
                out = layer.fc_self(h[b0:b1]) + neigh
                if l < len(self.layers) - 1:
                    out = self.dropout(self.activation(out))                 # :377-379 (dropout is the identity in eval mode)
                y[b0:b1] = out
            h = y
        g.ndata["h"] = h
        self.train(was_training)
        return h


def _full_neighbor_mean(g, h, b0, b1, dst_all):
    """mean over ALL in-neighbours of nodes b0..b1-1 (a MultiLayerFullNeighborSampler(1) block without edge weights,
    model.py:347-349, 375-376): the CSC columns of a contiguous node range are one contiguous slice."""
    import torch
    from .graph import Block
    from .nn import weighted_aggregate
    e0, e1 = int(g.indptr[b0]), int(g.indptr[b1])
    indptr = (g.indptr[b0:b1 + 1] - e0).to(torch.int32)
    src = g.indices[e0:e1]
    dst = dst_all[e0:e1] - b0
    blk = Block(None, h.shape[0], b1 - b0, indptr, src, dst, src, src, torch.empty(0, dtype=torch.int32, device=h.device))
    blk._transposed = (None, None)
    return weighted_aggregate(blk, h, None, mean=True)


class GCN(nn.Module):
    """model.py:386-439: stacked ``dglnn.GraphConv(norm='both', allow_zero_in_degree=True)`` with the sampler's edge
    weights.  (Unreachable from the reference's CLI -- ``--model gcn`` trains SAGE, train_lightning.py:597-607 -- kept
    for API completeness.)"""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout):
        super().__init__()
        from .nn import GraphConv
        self.n_layers, self.n_hidden, self.n_classes = n_layers, n_hidden, n_classes
        self.layers = nn.ModuleList()
        if n_layers > 1:
            self.layers.append(GraphConv(in_feats, n_hidden, activation=activation, allow_zero_in_degree=True))
            for _ in range(1, n_layers - 1):
                self.layers.append(GraphConv(n_hidden, n_hidden, activation=activation, allow_zero_in_degree=True))
            self.layers.append(GraphConv(n_hidden, n_classes, allow_zero_in_degree=True))
        else:
            self.layers.append(GraphConv(in_feats, n_classes, allow_zero_in_degree=True))
        self.dropout = nn.Dropout(dropout)
        self.activation = activation

    def forward(self, blocks, x):
        h, norm = x, _gathered_norm(blocks, x)
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            block.srcdata["embed_norm"] = embed_norm(h) if (l > 0 or norm is None) else norm      # model.py:425-427
            h = layer(block, h, edge_weight=(block.edata["edge_weights"] if "edge_weights" in block.edata else None))
            if l < len(self.layers) - 1:
                h = self.dropout(h)                                           # model.py:437-438
        return h

    @torch.no_grad()
    def inference(self, g, device=None, batch_size=128, use_uva=False, num_workers=0):
        """model.py:441-488.  GraphConv(norm='both') normalises by the BLOCK's out-degrees, so the result depends on how
        the nodes are batched: ``batch_size`` is honoured exactly (contiguous batches, unshuffled, last one short)."""
        was_training = self.training
        self.eval()
        n = len(self.layers)
        h = _blockwise_inference(self.layers, g, g.ndata["features"], int(batch_size),
                                 lambda l, out: self.dropout(out) if l < n - 1 else out)
        self.train(was_training)
        return h


def _blockwise_inference(layers, g, h, batch_size, finish):
    """The shared loop of the reference's three ``inference`` methods: per layer, walk all nodes in contiguous batches,
    run the layer on the full-neighbour block of the batch and write the rows into y (model.py:267-288, 472-487)."""
    from .graph import full_neighbor_block
    V = g.num_nodes()
    for l, layer in enumerate(layers):
        y = None
        for b0 in range(0, V, batch_size):
            b1 = min(V, b0 + batch_size)
            blk = full_neighbor_block(g, b0, b1)
            out = finish(l, layer(blk, h[blk.srcdata["_ID"].long()]))
            if y is None:
                y = torch.empty(V, out.shape[1], dtype=h.dtype, device=h.device)
            y[b0:b1] = out
        h = y
        g.ndata["h"] = h
    return h


class GATv2(nn.Module):
    """model.py:115-234.  ``forward`` stores embed_norm (srcdata) and the head-mean pre-softmax logits ``a_ij`` (edata)
    that the bandit's calculate_alpha consumes (model.py:211-213, 224-227)."""

    def __init__(self, num_layers, in_dim, num_hidden, num_classes, heads, activation, feat_drop, attn_drop, negative_slope,
                 residual):
        super().__init__()
        from .nn import GATv2Conv
        self.num_layers, self.activation = num_layers, activation
        self.num_hidden, self.num_classes, self.heads = num_hidden, num_classes, heads
        mk = lambda i, o, h, res, act: GATv2Conv(i, o, h, feat_drop, attn_drop, negative_slope, res, act, bias=False,
                                                 share_weights=True, allow_zero_in_degree=True)
        self.gatv2_layers = nn.ModuleList()
        if num_layers > 1:
            self.gatv2_layers.append(mk(in_dim, num_hidden, heads[0], False, activation))                     # :141-155
            for l in range(1, num_layers - 1):
                self.gatv2_layers.append(mk(num_hidden * heads[l - 1], num_hidden, heads[l], residual, activation))
            self.gatv2_layers.append(mk(num_hidden * heads[-2], num_classes, heads[-1], residual, None))       # :175-189
        else:
            self.gatv2_layers.append(mk(in_dim, num_classes, heads[-1], residual, None))
        for l, layer in enumerate(self.gatv2_layers):              # the attention-dropout stream of layer l (nn.GATv2Conv._fused_state)
            layer._drop_salt = l + 1

    def forward(self, blocks, inputs):
        h = inputs.bfloat16()
        norm = _gathered_norm(blocks, h)
        for l, block in enumerate(blocks):
            block.srcdata["embed_norm"] = embed_norm(h) if (l > 0 or norm is None) else norm                   # :211-213
            h, a = self.gatv2_layers[l](block, h, edge_weight=(block.edata["edge_weights"] if "edge_weights" in block.edata else None),
                                        get_attention=True)
            block.edata["a_ij"] = a.squeeze(-1).mean(dim=1)                                                    # :224-227
            h = h.flatten(1) if l < len(blocks) - 1 else h.mean(1)                                             # :228-232
        return h

    @torch.no_grad()
    def inference(self, g, device=None, batch_size=128, use_uva=False, num_workers=0, node_chunk=4096):
        """model.py:236-289.  Every output row depends only on its own in-edges (edge softmax is per destination), so the
        nodes are walked ``node_chunk`` at a time instead of the reference loader's ``batch_size``; the rows are the same."""
        was_training = self.training
        self.eval()                                                                                            # :265
        n = len(self.gatv2_layers)
        h = _blockwise_inference(self.gatv2_layers, g, g.ndata["features"].bfloat16(), int(node_chunk),
                                 lambda l, out: out.flatten(1) if l < n - 1 else out.mean(1))                  # :282-285
        self.train(was_training)
        return h
