"""GPU suite: the one-workgroup-per-destination GATv2 kernels (csrc/gat_fused.hip) against the separate kernels of csrc/gat.hip
(same bits forward), fp32 autograd (backward), and the attention-dropout stream's law."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ulp(a, b):
    def key(t):
        x = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32).numpy() & 0xFFFF
        return np.where(x & 0x8000, -(x & 0x7FFF), x & 0x7FFF)
    return np.abs(key(a) - key(b))


def _block(cuda, V=3000, E=50000, fan=300, n_seeds=60, seed=41):
    import bliss_gnn_amd as bg
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(V, E, seed=seed)
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda))
    s = bg.PoissonBanditLadiesSampler([fan], eta=0.1)
    torch.manual_seed(1)
    _, _, (blk,) = s.sample_blocks(g, torch.arange(n_seeds, dtype=torch.int32, device=cuda))
    return blk


@pytest.mark.parametrize("fin,H,D,residual", [(40, 4, 16, False), (64, 4, 256, True), (64, 1, 41, True), (48, 8, 32, False), (32, 2, 6, False)])
def test_fused_forward_has_the_bits_of_the_separate_kernels(cuda, monkeypatch, fin, H, D, residual):
    """logits, rst (and with them a_ij) of the fused launch == logits kernel + edge softmax + aggregation, bit for bit (both round
    where the reference's bf16 ops round; the fixtures of tests/golden pin either path), incl. head widths that are not a
    multiple of four (scalar path) and 8 heads."""
    from bliss_gnn_amd.nn import GATv2Conv
    blk = _block(cuda)
    K = blk.num_src_nodes()
    torch.manual_seed(5)
    layer = GATv2Conv(fin, D, H, 0.0, 0.0, 0.2, residual, None, bias=False, share_weights=True, allow_zero_in_degree=True).to(cuda).bfloat16()
    h = (torch.randn(K, fin, generator=torch.Generator().manual_seed(6)) * 0.5).bfloat16().to(cuda)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BLISS_GAT_FUSED", mode)
        hd = h.clone().requires_grad_(True)
        out, e = layer(blk, hd, get_attention=True)
        layer.zero_grad(set_to_none=True)
        gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).bfloat16().to(cuda)
        (out * gout).float().sum().backward()
        outs[mode] = (out.detach().clone(), e.detach().clone(), hd.grad.clone(), layer.fc_src.weight.grad.clone(), layer.attn.grad.clone())
    f, u = outs["1"], outs["0"]
    assert torch.equal(f[1].view(torch.int16), u[1].view(torch.int16)), "logits"
    # rst: the same fp32 products, added in another order (eight waves of a row vs chunks of 16 edges): a rounding tie may flip
    d = _ulp(f[0], u[0])
    assert d.max() <= 1 and (d > 0).mean() <= 0.02, (int(d.max()), float((d > 0).mean()))
    # the backward paths round at different points (one pass by source instead of two): a few bf16 ulps of each tensor's scale
    for name, a, b in (("d_h", f[2], u[2]), ("d_W", f[3], u[3]), ("d_attn", f[4], u[4])):
        assert (a.float() - b.float()).abs().max() <= 4 * 2.0 ** -8 * b.float().abs().max(), name


def _gat_ref(blk_src, blk_dst, S, h, W, attn, H, D, slope, mask_scale=None):
    fs = (h @ W.t()).view(-1, H, D)
    x = torch.nn.functional.leaky_relu(fs[blk_src] + fs[blk_dst], slope)
    e = (x * attn.view(1, H, D)).sum(-1)
    m = torch.full((S, H), -float("inf"), dtype=e.dtype).scatter_reduce(0, blk_dst[:, None].expand(-1, H), e.detach(), "amax")
    ex = torch.exp(e - m[blk_dst])
    a = ex / torch.zeros(S, H, dtype=e.dtype).index_add_(0, blk_dst, ex)[blk_dst]
    if mask_scale is not None:
        a = a * mask_scale
    return torch.zeros(S, H, D, dtype=e.dtype).index_add_(0, blk_dst, a[:, :, None] * fs[blk_src]), e


def test_fused_backward_vs_fp32_autograd_with_dropout(cuda):
    """Training mode with attention dropout 0.3: the mask the kernel drew is read back from its outputs (a_drop = a * mask /
    (1 - p)), then forward and all three gradients are compared with fp64 autograd of the same formulas under that mask;
    the keep rate follows Bernoulli(1 - p) and two launches draw different masks."""
    from bliss_gnn_amd.nn import GATv2Conv, _GatFusedMP
    blk = _block(cuda, fan=400, n_seeds=80)
    K, S = blk.num_src_nodes(), blk.num_dst_nodes()
    src, dst = blk.src.cpu().long(), blk.dst.cpu().long()
    H, D, fin, p = 4, 32, 48, 0.3
    torch.manual_seed(5)
    layer = GATv2Conv(fin, D, H, 0.0, p, 0.2, False, None, bias=False, share_weights=True, allow_zero_in_degree=True).to(cuda).bfloat16()
    layer.train()
    h = (torch.randn(K, fin, generator=torch.Generator().manual_seed(6)) * 0.5).bfloat16()
    hd = h.to(cuda).requires_grad_(True)
    feat = layer.fc_src(hd)
    st = layer._fused_state(cuda)
    rst, e = _GatFusedMP.apply(feat, layer.attn, blk, H, D, 0.2, p, st)
    node = rst.grad_fn
    a, ad = node.saved_tensors[2].float().cpu(), node.saved_tensors[3].float().cpu()
    keep = ad != 0
    nz = a > 2.0 ** -100
    rate = keep[nz].float().mean().item()
    n = int(nz.sum())
    assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 0.01, rate
    assert torch.equal(ad[keep & nz], (a[keep & nz] * (1 / (1 - p))).bfloat16().float())       # a * mask * scale in bf16 (nn.Dropout's arithmetic)
    mask_scale = keep.double() / (1 - p)
    gout = torch.randn(S, H * D, generator=torch.Generator().manual_seed(7)).bfloat16()
    (rst * gout.to(cuda)).float().sum().backward()
    W = layer.fc_src.weight.detach().double().cpu().requires_grad_(True)
    at = layer.attn.detach().double().cpu().requires_grad_(True)
    hr = h.double().requires_grad_(True)
    ref_out, ref_e = _gat_ref(src, dst, S, hr, W, at, H, D, 0.2, mask_scale)
    (ref_out.view(S, H * D) * gout.double()).sum().backward()
    tol = lambda t: 8 * t.abs().max() * 2 ** -8
    assert (e.float().cpu().double() - ref_e).abs().max() <= tol(ref_e)
    assert (rst.float().cpu().double() - ref_out.view(S, H * D)).abs().max() <= tol(ref_out)
    assert (hd.grad.float().cpu().double() - hr.grad).abs().max() <= 3 * tol(hr.grad)
    assert (layer.fc_src.weight.grad.float().cpu().double() - W.grad).abs().max() <= 3 * tol(W.grad)
    assert (layer.attn.grad.float().cpu().double().view(-1) - at.grad.view(-1)).abs().max() <= 3 * tol(at.grad)
    # the launch counter advanced: the next launch draws another mask
    with torch.no_grad():
        rst2, _ = _GatFusedMP.apply(feat.detach(), layer.attn.detach(), blk, H, D, 0.2, p, st)
    assert int(st["ctr"][0]) == 2 and int(st["ctr"][1]) == 0
    assert not torch.equal(rst2, rst.detach())


def test_fused_kernels_on_a_hub_destination(cuda):
    """A destination with thousands of in-edges (full-neighbour blocks of the inference path have them): the row's eight waves
    split its edges; same bits as the separate kernels."""
    import bliss_gnn_amd as bg
    from bliss_gnn_amd.graph import full_neighbor_block
    from bliss_gnn_amd.nn import GATv2Conv
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(4000, 200000, seed=9)
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda))
    deg = (ip[1:] - ip[:-1])
    hub = int(deg.argmax())
    b0 = max(0, hub - 3)
    blk = full_neighbor_block(g, b0, b0 + 8)
    assert int(deg[hub]) > 600
    torch.manual_seed(2)
    layer = GATv2Conv(24, 64, 4, 0.0, 0.0, 0.2, False, None, bias=False, share_weights=True, allow_zero_in_degree=True).to(cuda).bfloat16()
    h = (torch.randn(blk.num_src_nodes(), 24, generator=torch.Generator().manual_seed(3)) * 0.5).bfloat16().to(cuda)
    import os
    res = {}
    for mode in ("1", "0"):
        os.environ["BLISS_GAT_FUSED"] = mode
        with torch.no_grad():
            res[mode] = layer(blk, h, get_attention=True)
    os.environ.pop("BLISS_GAT_FUSED")
    assert torch.equal(res["1"][1].view(torch.int16), res["0"][1].view(torch.int16))
    d = _ulp(res["1"][0], res["0"][0])
    assert d.max() <= 1 and (d > 0).mean() <= 0.02
