"""GPU suite: the model side against fixtures produced by RUNNING the reference's own model.py (tests/golden/make_golden.py:
gat_case / sage_model_case) -- custom_GATv2Conv.forward + GATv2.forward + exp3(model='gat'), and SAGE.forward.
Nothing here reads /root/reference."""
import numpy as np
import pytest
import torch


from conftest import bf16_bits, bits_to_bf16, golden_cases, load_golden

pytestmark = pytest.mark.gpu


def _ulp(a_bits, b_bits):
    def key(x):
        x = x.astype(np.int32)
        return np.where(x & 0x8000, -(x & 0x7FFF), x & 0x7FFF)
    return np.abs(key(a_bits) - key(b_bits))


def embed_norm(h):
    from bliss_gnn_amd.nn import embed_norm as f
    return f(h)


def _setup(z, cuda, model):
    import bliss_gnn_amd as bg
    ip, ix, ei = torch.from_numpy(z["indptr"]), torch.from_numpy(z["indices"]), torch.from_numpy(z["eid"])
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda))
    g.edata["w"] = bg.normalized_edata(g)
    g.ndata["features"] = bits_to_bf16(z["features"]).to(cuda)
    s = bg.PoissonBanditLadiesSampler(z["fanouts"].tolist(), importance_sampling=1, node_embedding="features", num_steps=1000,
                                      eta=float(z["eta"]), model=model)
    return bg, g, s


def _check_block(bg, z, prefix, blk):
    assert np.array_equal(z[prefix + "src"], blk.src.cpu().numpy())
    assert np.array_equal(z[prefix + "dst"], blk.dst.cpu().numpy())
    assert np.array_equal(z[prefix + "eid"], blk.edata[bg.EID].cpu().numpy())
    assert np.array_equal(z[prefix + "src_nid"], blk.srcdata[bg.NID].cpu().numpy())
    assert np.array_equal(z[prefix + "edge_weights"], bf16_bits(blk.edata["edge_weights"]))
    assert np.array_equal(z[prefix + "q_ij"], bf16_bits(blk.edata["q_ij"]))
    assert np.array_equal(z[prefix + "node_prob"], bf16_bits(blk.srcdata["node_prob"]))


def _gat_layers(z, cuda, n_layers):
    from bliss_gnn_amd.nn import GATv2Conv
    heads, hidden, classes = z["heads"].tolist(), int(z["hidden"]), int(z["classes"])
    layers = []
    for l in range(n_layers):
        D = hidden if l < n_layers - 1 else classes
        fc = bits_to_bf16(z[f"p{l}_fc_src"])
        kind = int(z[f"p{l}_res_kind"])
        act = torch.nn.functional.elu if l < n_layers - 1 else None
        layer = GATv2Conv(fc.shape[1], D, heads[l], 0.0, 0.0, 0.2, kind != 0, act, bias=False, share_weights=True,
                          allow_zero_in_degree=True).to(cuda).bfloat16()
        with torch.no_grad():
            layer.fc_src.weight.copy_(fc.to(cuda))
            layer.attn.copy_(bits_to_bf16(z[f"p{l}_attn"]).view(1, heads[l], D).to(cuda))
            if kind == 1:
                layer.res_fc.weight.copy_(bits_to_bf16(z[f"p{l}_res_fc"]).to(cuda))
            else:
                assert (kind == 2) == isinstance(layer.res_fc, torch.nn.Identity)
        layers.append(layer)
    return layers


@pytest.mark.parametrize("name", golden_cases("gat"))
def test_gat_layers_and_exp3_vs_reference_run(cuda, name):
    """a19 / a13-GAT pinned to a run of the reference's own Python.  Per layer, from the fixture's layer input: the logits e
    (what the reference returns as attention), the layer output rst, a_ij and embed_norm within the stated bf16 ulps of the
    reference's bf16 CPU run (the kernels round where its tensor ops round; what is left is the fp32 summation order inside
    the Linear / the sums); then the bandit update from the FIXTURE's a_ij / embed_norm: alpha-dependent rewards and the
    EXP3 rows bit for bit over two consecutive steps."""
    z = load_golden(name)
    bg, g, s = _setup(z, cuda, "gat")
    fanouts, seed = z["fanouts"].tolist(), int(z["torch_seed"])
    layers = _gat_layers(z, cuda, len(fanouts))
    worst = dict(e=0, rst=0, a_ij=0, en=0)
    for step in range(int(z["n_steps"])):
        seeds = torch.from_numpy(z[f"s{step}_seeds"]).to(cuda)
        torch.manual_seed(seed + 10 + step)
        inp, _, blocks = s.sample_blocks(g, seeds)
        for l, blk in enumerate(blocks):
            pre = f"s{step}_l{l}_"
            _check_block(bg, z, pre, blk)
            h_in = bits_to_bf16(z[pre + "h_in"]).to(cuda)
            with torch.no_grad():
                rst, e = layers[l](blk, h_in, get_attention=True)
                a_ij = e.squeeze(-1).mean(dim=1)                                   # model.py:224-227
                en = embed_norm(h_in)
            d_e = _ulp(bf16_bits(e.reshape(e.shape[0], -1)), z[pre + "e"])
            d_r = _ulp(bf16_bits(rst), z[pre + "rst"])
            d_a = _ulp(bf16_bits(a_ij), z[pre + "a_ij"])
            d_n = _ulp(bf16_bits(en), z[pre + "embed_norm"])
            worst = dict(e=max(worst["e"], int(d_e.max())), rst=max(worst["rst"], int(d_r.max())),
                         a_ij=max(worst["a_ij"], int(d_a.max())), en=max(worst["en"], int(d_n.max())))
            # logits: a sum of ~D rounded terms in fp32 -- the order of the additions is all that differs
            assert d_e.max() <= 1 and (d_e > 0).mean() <= 0.02, (pre, int(d_e.max()), float((d_e > 0).mean()))
            assert d_n.max() <= 1
            # rst: softmax weights from logits that may differ by an ulp, summed in fp32, then residual + elu in bf16:
            # <= 2 ulps except where the output cancels to a small value -- bounded in absolute terms there
            ref = bits_to_bf16(z[pre + "rst"]).float()
            got = rst.float().cpu()
            scale = ref.abs().max()
            bad = (d_r > 2) & ((got - ref).abs().numpy() > float(scale) * 2 ** -9)
            assert not bad.any(), (pre, int(d_r.max()), int(bad.sum()))
            assert (d_a > 1).mean() <= 0.01, (pre, int(d_a.max()))
            # the bandit update consumes the reference's own a_ij / embed_norm: bit-exact from here on
            blk.edata["a_ij"] = bits_to_bf16(z[pre + "a_ij"]).to(cuda)
            blk.srcdata["embed_norm"] = bits_to_bf16(z[pre + "embed_norm"]).to(cuda)
        s.exp3(blocks, g)
        s.check_errors()
        for l, blk in enumerate(blocks):
            assert np.array_equal(z[f"s{step}_l{l}_rewards"], bf16_bits(blk.edata["rewards"]))
        assert np.array_equal(z[f"s{step}_exp3_weights"], bf16_bits(s.exp3_weights))
    print("worst ulp distances", worst)


@pytest.mark.parametrize("name", golden_cases("gat"))
def test_gat_model_forward_vs_reference_run(cuda, name):
    """GATv2.forward end to end (model.py:207-234) with the fixture's parameters: class logits within bf16 rounding noise of
    the reference run after three layers (each layer's tolerance is checked layer by layer above)."""
    from bliss_gnn_amd.model import GATv2
    z = load_golden(name)
    bg, g, s = _setup(z, cuda, "gat")
    fanouts, seed = z["fanouts"].tolist(), int(z["torch_seed"])
    F = z["features"].shape[1]
    net = GATv2(len(fanouts), F, int(z["hidden"]), int(z["classes"]), z["heads"].tolist(), torch.nn.functional.elu, 0.0, 0.0, 0.2,
                bool(int(z["residual"]))).to(cuda).bfloat16()
    for l, (mine, fix) in enumerate(zip(net.gatv2_layers, _gat_layers(z, cuda, len(fanouts)))):
        mine.load_state_dict(fix.state_dict())
    torch.manual_seed(seed + 10)
    inp, _, blocks = s.sample_blocks(g, torch.from_numpy(z["s0_seeds"]).to(cuda))
    with torch.no_grad():
        pred = net(blocks, blocks[0].srcdata["features"])
    ref = bits_to_bf16(z["s0_pred"]).float()
    assert (pred.float().cpu() - ref).abs().max() <= 4 * ref.abs().max() * 2 ** -8
    for l, blk in enumerate(blocks):
        assert (_ulp(bf16_bits(blk.srcdata["embed_norm"]), z[f"s0_l{l}_embed_norm"]) > 1).mean() <= (0.0 if l == 0 else 0.05)


def test_gat_fp32_mode_within_1e4(cuda):
    """North star: activations within 1e-4 rel.  The F32 variants of the three forward kernels (no intermediate rounding,
    float results) against fp32 torch math on the SAME bf16 operands, on a fixture block."""
    from bliss_gnn_amd.nn import gat_forward_f32
    z = load_golden("gat0_model_exp3")
    bg, g, s = _setup(z, cuda, "gat")
    torch.manual_seed(int(z["torch_seed"]) + 10)
    _, _, blocks = s.sample_blocks(g, torch.from_numpy(z["s0_seeds"]).to(cuda))
    gen = torch.Generator().manual_seed(3)
    for blk, (H, D) in zip(blocks, ((4, 16), (4, 64), (1, 40))):
        K, S = blk.num_src_nodes(), blk.num_dst_nodes()
        feat = (torch.randn(K, H * D, generator=gen) * 0.7).bfloat16()
        attn = (torch.randn(H * D, generator=gen) * 0.3).bfloat16()
        e, a, out = gat_forward_f32(blk, feat.to(cuda), attn.to(cuda), H, D, 0.2)
        src, dst = blk.src.cpu().long(), blk.dst.cpu().long()
        f = feat.double().view(K, H, D)
        x = torch.nn.functional.leaky_relu(f[src] + f[dst], 0.2)
        e_ref = (x * attn.double().view(1, H, D)).sum(-1)
        m = torch.full((S, H), -float("inf"), dtype=torch.float64).scatter_reduce(0, dst[:, None].expand(-1, H), e_ref, "amax")
        ex = torch.exp(e_ref - m[dst])
        a_ref = ex / torch.zeros(S, H, dtype=torch.float64).index_add_(0, dst, ex)[dst]
        o_ref = torch.zeros(S, H, D, dtype=torch.float64).index_add_(0, dst, a_ref[:, :, None] * f[src]).view(S, H * D)
        rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert rel(e, e_ref) <= 1e-4 and rel(a, a_ref) <= 1e-4 and rel(out, o_ref) <= 1e-4, (rel(e, e_ref), rel(a, a_ref), rel(out, o_ref))


def test_sage_model_vs_reference_run(cuda):
    """a17: SAGE.forward with the fixture's parameters on the HIP sampler's blocks (== the reference run's, bit for bit): per
    layer from the fixture's layer input, outputs within bf16 rounding of the reference's bf16 CPU layer, embed_norm within
    one ulp; then exp3 from the fixture's norms bit for bit."""
    from bliss_gnn_amd.model import SAGE
    z = load_golden("sage0_model_exp3")
    bg, g, s = _setup(z, cuda, "sage")
    fanouts, seed = z["fanouts"].tolist(), int(z["torch_seed"])
    F = z["features"].shape[1]
    net = SAGE(F, int(z["hidden"]), int(z["classes"]), len(fanouts), torch.relu, 0.0).to(cuda).bfloat16()
    with torch.no_grad():
        for l, layer in enumerate(net.layers):
            layer.fc_neigh.weight.copy_(bits_to_bf16(z[f"p{l}_w_neigh"]).to(cuda))
            layer.fc_self.weight.copy_(bits_to_bf16(z[f"p{l}_w_self"]).to(cuda))
            layer.fc_self.bias.copy_(bits_to_bf16(z[f"p{l}_b_self"]).to(cuda))
    torch.manual_seed(seed + 10)
    inp, _, blocks = s.sample_blocks(g, torch.from_numpy(z["seeds"]).to(cuda))
    for l, blk in enumerate(blocks):
        _check_block(bg, z, f"l{l}_", blk)
    net.eval()
    with torch.no_grad():
        pred = net(blocks, blocks[0].srcdata["features"])
        ref = bits_to_bf16(z["pred"]).float()
        assert (pred.float().cpu() - ref).abs().max() <= 4 * ref.abs().max() * 2 ** -8
        for l, (layer, blk) in enumerate(zip(net.layers, blocks)):
            h_in = bits_to_bf16(z[f"l{l}_h_in"]).to(cuda)
            out = layer(blk, h_in, edge_weight=blk.edata["edge_weights"]).float().cpu()
            ref = bits_to_bf16(z[f"l{l}_out"]).float()
            assert (out - ref).abs().max() <= 2 * ref.abs().max() * 2 ** -8, l
            assert _ulp(bf16_bits(embed_norm(h_in)), z[f"l{l}_embed_norm"]).max() <= 1
            blk.srcdata["embed_norm"] = bits_to_bf16(z[f"l{l}_embed_norm"]).to(cuda)
    s.exp3(blocks, g)
    s.check_errors()
    for l, blk in enumerate(blocks):
        assert np.array_equal(z[f"l{l}_rewards"], bf16_bits(blk.edata["rewards"]))
    assert np.array_equal(z["exp3_weights"], bf16_bits(s.exp3_weights))


def _row_checksum(w):
    b = bf16_bits(w).astype(np.uint64).reshape(-1)
    idx = np.arange(1, b.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.uint64(((b + np.uint64(1)) * (idx * np.uint64(0x9E3779B97F4A7C15))).sum())


def test_collapse_on_normal_features_hits_the_references_step(cuda):
    """The non-finite error the sampler raises on N(0,1) features after a few hundred steps is the REFERENCE's own end, at the
    reference's own step: a reference run (tests/golden/make_golden.py:collapse_case) dies in torch.bernoulli at step 616 when a
    seed column's weights have all underflowed to zero; the HIP path produces the reference's EXP3 rows after every one of
    the 616 updates before it (checksums; subnormal column sums included) and raises at exactly that step."""
    import bliss_gnn_amd as bg
    z = load_golden("collapse0_normal_features")
    ip, ix, ei = torch.from_numpy(z["indptr"]), torch.from_numpy(z["indices"]), torch.from_numpy(z["eid"])
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda))
    g.edata["w"] = bg.normalized_edata(g)
    fanouts, eta, seed, batch = z["fanouts"].tolist(), float(z["eta"]), int(z["torch_seed"]), int(z["batch"])
    T, V = int(z["failed_at"]), ip.numel() - 1
    norm0, norm1 = bits_to_bf16(z["norm0"]).to(cuda), bits_to_bf16(z["norm1"]).to(cuda)
    s = bg.PoissonBanditLadiesSampler(fanouts, importance_sampling=1, node_embedding="features", num_steps=1000, eta=eta, model="sage")
    gen = torch.Generator().manual_seed(seed + 3)
    for step in range(T + 1):
        seeds = torch.randperm(V, generator=gen)[:batch].to(torch.int32).to(cuda)
        torch.manual_seed(seed + 1000 + step)
        if step == T:
            with pytest.raises(RuntimeError):
                s.sample_blocks(g, seeds)
                s.check_errors()
            break
        _, _, blocks = s.sample_blocks(g, seeds)
        s.check_errors()
        assert [b.num_src_nodes() for b in blocks] == z["kept"][step].tolist(), step
        for l, b in enumerate(blocks):
            b.srcdata["embed_norm"] = (norm0 if l == 0 else norm1)[b.srcdata[bg.NID].long()]
        s.exp3(blocks, g)
        s.check_errors()
        assert _row_checksum(s.exp3_weights) == z["checksums"][step], f"EXP3 rows left the reference's trajectory at step {step}"
    assert np.array_equal(bf16_bits(s.exp3_weights), z["last_weights"])
