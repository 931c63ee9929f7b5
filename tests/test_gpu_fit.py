"""GPU suite: the experiment protocol (bliss_gnn_amd/fit.py) end to end on a small learnable synthetic task, for every
``--sampler`` name of train_lightning.py:536-540."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _task(cuda, V=3000, E=40000, F=24, classes=4):
    import bliss_gnn_amd as bg
    from bliss_gnn_amd.synth import chung_lu_csc
    ip, ix, ei = chung_lu_csc(V, E, seed=21)
    gen = torch.Generator().manual_seed(2)
    feats = torch.randn(V, F, generator=gen).bfloat16()
    labels = (feats.float() @ torch.randn(F, classes, generator=gen)).argmax(1)
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
    perm = torch.randperm(V, generator=gen).to(torch.int32).to(cuda)
    return g, perm[:1800], perm[1800:2300], perm[2300:]


@pytest.mark.parametrize("name", ["poisson-bandit", "bandit", "poisson-ladies", "ladies", "neighbor", "full"])
def test_fit_runs_every_sampler_choice(cuda, name):
    from bliss_gnn_amd import fit
    from bliss_gnn_amd.model import SAGE
    g, tr, va, te = _task(cuda)
    if name not in ("neighbor", "full"):
        import bliss_gnn_amd as bg
        g.edata["w"] = bg.normalized_edata(g)                                     # train_lightning.py:359,362
    sampler = fit.make_sampler(name, [64, 32, 16])
    torch.manual_seed(0)
    model = SAGE(24, 32, 4, 3, torch.relu, 0.1).to(cuda).bfloat16()
    seen = []
    out = fit.fit(g, sampler, model, tr, va, te, batch_size=128, lr=0.01, max_epochs=4, log=seen.append)
    assert len(out["history"]) == 4 and out["steps"] == 4 * (1800 // 128)
    assert out["history"][-1]["train_loss"] < out["history"][0]["train_loss"]     # it learns
    assert out["best_val_acc"] > 0.3 and set(out["final"]) == {"Train", "Validation", "Test"}
    assert out["final"]["Test"] > 0.3                                             # 4 classes: chance is 0.25
    assert seen == out["history"]


def test_full_neighbor_blocks_are_the_whole_neighbourhood(cuda):
    from bliss_gnn_amd import fit
    g, tr, _, _ = _task(cuda)
    s = fit.MultiLayerFullNeighborSampler(2)
    _, _, blocks = s.sample(g, tr[:50])
    for b in blocks:
        dst = b.dstdata["_ID"].long()
        assert torch.equal(b.in_degrees().long(), (g.indptr[dst + 1] - g.indptr[dst]))      # every in-edge of every seed
        assert bool((b.edata["edge_weights"] == 1).all())


def test_k_runs_and_lr_schedule_inside_fit(cuda):
    from bliss_gnn_amd import fit
    from bliss_gnn_amd.model import SAGE
    g, tr, va, te = _task(cuda)
    import bliss_gnn_amd as bg
    g.edata["w"] = bg.normalized_edata(g)

    def one(i):
        torch.manual_seed(i)
        model = SAGE(24, 16, 4, 3, torch.relu, 0.0).to(cuda).bfloat16()
        return fit.fit(g, fit.make_sampler("poisson-bandit", [32, 16, 8]), model, tr, va, te, batch_size=256, lr=0.01, max_epochs=7, seed=i)

    res = fit.k_runs(one, 2)
    lrs = [h["lr"] for h in res["runs"][0]["history"]]
    assert lrs[:5] == [0.01] * 5 and all(abs(x - 1e-4) < 1e-12 for x in lrs[5:])  # x 0.01 after 5 epochs (train_lightning.py:208)
    assert res["reduced"]["Test"]["n"] == 2 and 0 <= res["reduced"]["Test"]["std"] < 0.5
