"""GPU suite: the drop-in boundary -- what the reference's own call sites hand over.

* ``dgl.dataloading.DataLoader(g, ...)`` calls ``sampler.sample(g, ids)`` with a DGLGraph and the callback calls
  ``sampler.exp3(mfgs, g)`` (train_lightning.py:396-408, 469-471): a minimal fake exposing ONLY adj_tensors('csc'), ndata,
  edata, num_nodes() must give exactly the blocks and EXP3 rows of the same graph handed over as a bliss Graph.
* the reference's custom_GATv2Conv.forward (model.py:63-110) is written against graph.srcdata/dstdata/edata,
  apply_edges(fn.u_add_v), edge_softmax and update_all(fn.u_mul_e, fn.sum): that call sequence on a bliss Block must give
  the fused layer's result, forward and backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class FakeDGLGraph:
    """Nothing but the four members graph.as_graph reads."""

    def __init__(self, indptr, indices, eids, ndata, edata):
        self._csc, self.ndata, self.edata = (indptr, indices, eids), ndata, edata

    def adj_tensors(self, fmt):
        assert fmt == "csc"
        return self._csc

    def num_nodes(self):
        return self._csc[0].numel() - 1


def test_sampler_accepts_a_dgl_like_graph(cuda):
    import bliss_gnn_amd as bg
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(5000, 80000, seed=3)
    feats = torch.randn(5000, 16, generator=torch.Generator().manual_seed(1)).bfloat16().to(cuda)
    native = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats})
    native.edata["w"] = bg.normalized_edata(native)
    fake = FakeDGLGraph(ip.to(cuda), ix.to(cuda).long(), ei.to(cuda).long(), {"features": feats}, {"w": native.edata["w"]})   # int64 ids, as DGL may hand them
    seeds = torch.randperm(5000, generator=torch.Generator().manual_seed(2))[:64].to(torch.int32).to(cuda)
    outs = []
    for g in (native, fake):
        s = bg.PoissonBanditLadiesSampler([256, 128, 64], importance_sampling=1, node_embedding="features", eta=0.1, model="sage")
        rows = []
        for step in range(2):
            torch.manual_seed(step)
            inp, outp, blocks = s.sample(g, seeds)                                   # BlockSampler.sample, as the DataLoader calls it
            assert torch.equal(blocks[0].srcdata["features"], feats[inp.long()])    # train_lightning.py:138
            for b in blocks:
                b.srcdata["embed_norm"] = (torch.arange(b.num_src_nodes(), device=cuda) % 17 + 1).bfloat16()
            s.exp3(blocks, g)                                                        # train_lightning.py:471
            rows.append([(b.src.clone(), b.edata[bg.EID].clone(), b.edata["edge_weights"].view(torch.int16).clone()) for b in blocks])
        s.check_errors()
        outs.append((rows, s.exp3_weights.view(torch.int16).clone()))
        assert s._engine.g is (g if g is native else s._graph(g)) and s._graph(g) is s._graph(g)     # converted once, cached
    for ra, rb in zip(outs[0][0], outs[1][0]):
        for a, b in zip(ra, rb):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(TypeError):
        bg.PoissonBanditLadiesSampler([4]).sample(object(), seeds)


def test_reference_gat_call_sequence_on_a_block(cuda):
    import bliss_gnn_amd as bg
    from bliss_gnn_amd import function as fn
    from bliss_gnn_amd.nn import GATv2Conv, edge_softmax
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(3000, 50000, seed=9)
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda))
    g.edata["w"] = bg.normalized_edata(g)
    torch.manual_seed(0)
    _, _, blocks = bg.PoissonBanditLadiesSampler([200, 100]).sample_blocks(g, torch.arange(40, dtype=torch.int32, device=cuda))
    graph = blocks[0]
    H, D, IN = 4, 16, 24
    layer = GATv2Conv(IN, D, H, 0.0, 0.0, 0.2, True, None, bias=False, share_weights=True, allow_zero_in_degree=True).to(cuda).bfloat16()
    feat = torch.randn(graph.num_src_nodes(), IN, generator=torch.Generator().manual_seed(4)).bfloat16().to(cuda)
    gout = torch.randn(graph.num_dst_nodes(), H, D, generator=torch.Generator().manual_seed(5)).bfloat16().to(cuda)

    def reference_sequence(x):
        """model.py:63-110 (share_weights, block input), line by line on the DGL surface."""
        with graph.local_scope():                                                     # :48
            h_src = h_dst = x                                                         # :69 (feat_drop = 0)
            feat_src = layer.fc_src(h_src).view(-1, H, D)                             # :70
            feat_dst = feat_src[: graph.number_of_dst_nodes()]                        # :72-79
            h_dst = h_dst[: graph.number_of_dst_nodes()]
            graph.srcdata.update({"el": feat_src})                                    # :80
            graph.dstdata.update({"er": feat_dst})                                    # :81
            graph.apply_edges(fn.u_add_v("el", "er", "e"))                            # :82
            e = torch.nn.functional.leaky_relu(graph.edata.pop("e"), 0.2)            # :83-85
            e = (e * layer.attn).sum(dim=-1).unsqueeze(dim=2)                         # :86
            graph.edata["a"] = edge_softmax(graph, e)                                 # :88-90
            graph.update_all(fn.u_mul_e("el", "a", "m"), fn.sum("m", "ft"))           # :98
            rst = graph.dstdata["ft"]                                                 # :99
            rst = rst + layer.res_fc(h_dst).view(h_dst.shape[0], -1, D)               # :101-103
            return rst, e

    xa = feat.clone().requires_grad_()
    ra, ea = reference_sequence(xa)
    ra.backward(gout)
    ga = [p.grad.clone() for p in layer.parameters()]
    for p in layer.parameters():
        p.grad = None
    xb = feat.clone().requires_grad_()
    rb, eb = layer(graph, xb, get_attention=True)
    rb.backward(gout)
    gb = [p.grad.clone() for p in layer.parameters()]
    tol = dict(rtol=4e-2, atol=4e-2)                                                  # bf16 edge tensors vs the fused fp32-accumulating kernels
    assert "el" not in graph.srcdata and "a" not in graph.edata                       # local_scope restored the frames
    assert torch.allclose(ra.float(), rb.float(), **tol) and torch.allclose(ea.float(), eb.float(), **tol)
    assert torch.allclose(xa.grad.float(), xb.grad.float(), rtol=6e-2, atol=6e-2)
    for a, b in zip(ga, gb):
        assert torch.allclose(a.float(), b.float(), rtol=8e-2, atol=8e-2 * float(b.float().abs().max()))


def test_custom_ops_are_registered_and_traceable(cuda):
    """torch.ops.bliss.spmm / embed_norm: schema, fake (meta) kernels and the registered autograd formula checked by
    torch.library.opcheck; under FakeTensorMode the ops propagate shapes without touching the GPU library."""
    import bliss_gnn_amd as bg
    from bliss_gnn_amd import ops
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(2000, 30000, seed=4)
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda))
    g.edata["w"] = bg.normalized_edata(g)
    torch.manual_seed(1)
    _, _, (blk,) = bg.PoissonBanditLadiesSampler([150]).sample_blocks(g, torch.arange(30, dtype=torch.int32, device=cuda))
    h = torch.randn(blk.num_src_nodes(), 64, device=cuda).bfloat16().requires_grad_()
    t_indptr, t_edge = blk.transposed()
    args = (blk.indptr, blk.src, blk.dst, blk.edata["edge_weights"], h, blk.num_dst_nodes(), None, True, False, t_indptr, t_edge)
    torch.library.opcheck(torch.ops.bliss.spmm.default, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.bliss.embed_norm.default, (h.detach(),), test_utils=("test_schema", "test_faketensor"))
    out = torch.ops.bliss.spmm(*args)
    out.float().sum().backward()
    # reference: fp32 mean of weighted rows
    w = blk.edata["edge_weights"].float()
    ref = torch.zeros(blk.num_dst_nodes(), 64, device=cuda).index_add_(0, blk.dst.long(), h.detach().float()[blk.src.long()] * w[:, None])
    ref = ref / blk.in_degrees().clamp(min=1).float()[:, None]
    assert torch.allclose(out.float(), ref, rtol=2 ** -7, atol=2 ** -7 * float(ref.abs().max()))
    assert h.grad is not None and bool(torch.isfinite(h.grad.float()).all())
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode(allow_non_fake_inputs=True):
        fh = torch.empty(blk.num_src_nodes(), 64, dtype=torch.bfloat16, device=cuda)
        fo = torch.ops.bliss.spmm(blk.indptr, blk.src, blk.dst, None, fh, blk.num_dst_nodes(), None, True, True, None, None)
        assert fo.shape == (blk.num_dst_nodes(), 64) and fo.dtype == torch.float32
        assert torch.ops.bliss.embed_norm(fh).shape == (blk.num_src_nodes(),)
