"""GPU suite: the hand-written MFMA backward of the SAGE layers' Linears (csrc/sage_bwd.hip) -- what autograd derives for
fc_neigh / fc_self of dglnn.SAGEConv (model.py:303-308, 321-329) -- against exact integer data (operand layouts, transposed
LDS reads) and fp32 torch math (one bf16 rounding of an fp32-accumulated sum: <= 1 bf16 ulp)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, lo, hi, seed):
    return torch.randint(lo, hi + 1, shape, generator=torch.Generator().manual_seed(seed)).float().bfloat16()


@pytest.mark.parametrize("M,K1,N,K2,M2", [(100, 48, 256, 0, 0), (777, 256, 256, 256, 300), (65, 41, 256, 41, 32), (300, 256, 602, 0, 0),
                                           (33, 16, 40, 200, 33), (1, 7, 3, 0, 0)])
def test_dgrad_exact_integers(cuda, M, K1, N, K2, M2):
    """out = A1 W1 (+ A2 W2 on the first M2 rows) with small integers: every product and sum is exact in fp32 and the results
    are exact in bf16, so any slip in the fragment layouts or the transposed LDS reads shows as a wrong integer.  Asymmetric
    data, odd sizes, row counts from the device (capacity padding rows must come out as zeros)."""
    from bliss_gnn_amd.nn import sage_dgrad
    a1, w1 = _ints((M + 5, K1), -1, 1, 1), _ints((K1, N), -2, 2, 2)
    a2 = w2 = None
    if K2:
        a2, w2 = _ints((M2 + 3, K2), -1, 1, 3), _ints((K2, N), -2, 1, 4)
    ref = a1[:M].float() @ w1.float()
    if K2:
        ref[:M2] += a2[:M2].float() @ w2.float()
    assert ref.abs().max() <= 256
    m_dev = torch.tensor([M, M2], dtype=torch.int32, device=cuda)
    out = sage_dgrad(a1.to(cuda), w1.to(cuda), M + 5, m_dev.data_ptr(), a2=None if a2 is None else a2.to(cuda),
                     w2=None if w2 is None else w2.to(cuda), m2_bound=M2 + 3 if K2 else 0, m2_dev=m_dev.data_ptr() + 4 if K2 else 0)
    torch.cuda.synchronize()
    assert torch.equal(out[:M].float().cpu(), ref)
    assert not out[M:].any()


@pytest.mark.parametrize("R,n_out,k_in", [(200, 256, 128), (190, 41, 256), (70, 256, 602), (33, 7, 5), (129, 200, 130)])
def test_wgrad_exact_integers(cuda, R, n_out, k_in):
    """dW = D^T X and db = column sums of D with entries in {-1, 0, 1} and <= 200 rows: exact in bf16.  Two problems in one
    launch pair (the second without bias and with its own row count), rows beyond the device-side count ignored."""
    from bliss_gnn_amd.nn import sage_wgrad
    d, x = _ints((R + 40, n_out), -1, 1, 5), _ints((R + 40, k_in), -1, 1, 6)
    d2, x2 = _ints((R, n_out), -1, 1, 7), _ints((R + 9, k_in), -1, 1, 8)
    R2 = max(1, R - 17)
    cnt = torch.tensor([R, R2], dtype=torch.int32, device=cuda)
    (dw, db), (dw2, db2) = sage_wgrad([(d.to(cuda), x.to(cuda), R + 40, cnt.data_ptr(), True),
                                       (d2.to(cuda), x2.to(cuda), R, cnt.data_ptr() + 4, False)])
    torch.cuda.synchronize()
    assert db2 is None
    assert torch.equal(dw.float().cpu(), d[:R].float().t() @ x[:R].float())
    assert torch.equal(db.float().cpu(), d[:R].float().sum(0))
    assert torch.equal(dw2.float().cpu(), d2[:R2].float().t() @ x2[:R2].float())


@pytest.mark.parametrize("R,n_out,k_in,split", [(11000, 256, 602, None), (5000, 256, 602, "37"), (2100, 256, 256, None), (2000, 41, 256, None)])
def test_wgrad_vs_fp32_full_size(cuda, R, n_out, k_in, split, monkeypatch):
    """The weight-gradient shapes of the Reddit-like step (11 K x 256 x 602, ...) on random bf16 data against fp64 sums:
    within one bf16 rounding of the exact value (fp32 partial tiles summed in chunk order), and bitwise reproducible."""
    from bliss_gnn_amd.nn import sage_wgrad
    if split:
        monkeypatch.setenv("BLISS_WGRAD_WGS", split)       # (read once per process: only effective if this test runs first)
    g = torch.Generator().manual_seed(R)
    d = (torch.randn(R, n_out, generator=g) * 0.05).bfloat16()
    x = torch.randn(R, k_in, generator=g).bfloat16()
    Rt = R - 123
    cnt = torch.tensor([Rt], dtype=torch.int32, device=cuda)
    dd, xd = d.to(cuda), x.to(cuda)
    (dw, db), = sage_wgrad([(dd, xd, R, cnt.data_ptr(), True)])
    (dw_b, db_b), = sage_wgrad([(dd, xd, R, cnt.data_ptr(), True)])
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_b) and torch.equal(db, db_b)
    ref = (d[:Rt].double().t() @ x[:Rt].double())
    refb = d[:Rt].double().sum(0)
    err = (dw.double().cpu() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max() * 2.0 ** -8).all(), float((err / ref.abs().clamp_min(1e-9)).max())
    assert ((db.double().cpu() - refb).abs() <= refb.abs() * 2.0 ** -8 + 1e-6).all()


def test_dgrad_vs_fp32_full_size(cuda):
    from bliss_gnn_amd.nn import sage_dgrad
    g = torch.Generator().manual_seed(3)
    M, M2 = 3300, 1300
    a1, a2 = (torch.randn(M, 256, generator=g) * 0.1).bfloat16(), (torch.randn(M2, 256, generator=g) * 0.1).bfloat16()
    w1, w2 = (torch.randn(256, 256, generator=g) * 0.2).bfloat16(), (torch.randn(256, 256, generator=g) * 0.2).bfloat16()
    out = sage_dgrad(a1.to(cuda), w1.to(cuda), M, 0, a2=a2.to(cuda), w2=w2.to(cuda), m2_bound=M2, m2_dev=0)
    ref = a1.double() @ w1.double()
    ref[:M2] += a2.double() @ w2.double()
    err = (out.double().cpu() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max() * 2.0 ** -8).all()


def test_sage_training_gradients_match_the_library_path(cuda, monkeypatch):
    """One training step of the 3-layer SAGE (602 -> 256 -> 256 -> 41 shapes in miniature and at the hidden width the kernels
    tile for) with the MFMA backward against the same step with BLISS_SAGE_MFMA_BWD=0 (library GEMMs, round 2): every parameter
    gradient within two bf16 ulps of the gradient tensor's scale (both are fp32-accumulated, rounded at different points)."""
    import bliss_gnn_amd as bg
    from bliss_gnn_amd.model import SAGE
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(6000, 120000, seed=12)
    feats = torch.randn(6000, 300, generator=torch.Generator().manual_seed(1)).bfloat16()
    labels = torch.randint(0, 41, (6000,), generator=torch.Generator().manual_seed(2))
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
    g.edata["w"] = bg.normalized_edata(g)
    s = bg.PoissonBanditLadiesSampler([600, 300, 150], eta=0.1)
    torch.manual_seed(0)
    model = SAGE(300, 256, 41, 3, torch.relu, 0.0).to(cuda).bfloat16()
    torch.manual_seed(4)
    _, _, blocks = s.sample_blocks(g, torch.arange(128, dtype=torch.int32, device=cuda))
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BLISS_SAGE_MFMA_BWD", mode)
        model.zero_grad(set_to_none=True)
        pred = model(blocks, blocks[0].srcdata["features"])
        loss = torch.nn.functional.cross_entropy(pred.float(), blocks[-1].dstdata["labels"])
        loss.backward()
        grads[mode] = [p.grad.float().clone() for p in model.parameters()]
    for (name, _), a, b in zip(model.named_parameters(), grads["1"], grads["0"]):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 2 * 2.0 ** -8 * b.abs().max(), name
