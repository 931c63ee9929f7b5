"""CPU suite: the host-side control logic of bliss_gnn_amd/fit.py (train_lightning.py:205-216, 620-634, 711-733, 68-70)."""
import math

import torch

from bliss_gnn_amd import fit


class _Opt:
    def __init__(self, lr):
        self.param_groups = [dict(lr=lr)]


def test_steplr_per_epoch_matches_torch():
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.Adam([p], lr=0.002)
    ref = torch.optim.lr_scheduler.StepLR(ref_opt, gamma=0.01, step_size=5)           # the reference's scheduler (:208)
    mine_opt = _Opt(0.002)
    mine = fit.StepLR(mine_opt, 5, 0.01)
    for epoch in range(17):
        assert math.isclose(mine_opt.param_groups[0]["lr"], ref_opt.param_groups[0]["lr"], rel_tol=1e-12)
        ref_opt.step(); ref.step(); mine.step()


def test_early_stopping_threshold_and_patience():
    es = fit.EarlyStopping(stopping_threshold=0.9, patience=3)
    assert [es.should_stop(v) for v in (0.1, 0.2, 0.15, 0.18, 0.19)] == [False, False, False, False, True]   # 3 epochs without a new best
    es = fit.EarlyStopping(stopping_threshold=0.9, patience=1000)
    assert [es.should_stop(v) for v in (0.5, 0.9, 0.95)] == [False, False, True]                              # strictly above the target (Lightning: mode='max')


def test_checkpoint_keeps_the_best_parameters(tmp_path):
    m = torch.nn.Linear(3, 2)
    ck = fit.ModelCheckpoint(str(tmp_path / "best.pt"))
    with torch.no_grad():
        m.weight.fill_(1.0)
    assert ck.update(0.5, m)
    with torch.no_grad():
        m.weight.fill_(2.0)
    assert not ck.update(0.4, m)                                  # worse: not saved
    ck.restore(m)
    assert bool((m.weight == 1.0).all()) and ck.best == 0.5
    # on disk: the layout the reference's reload path reads (train_lightning.py:64, :671-682) -- 'state_dict' with the
    # 'module.' prefix of ModelLightning.module -- readable by the safe loader
    raw = torch.load(str(tmp_path / "best.pt"), weights_only=True)
    assert set(raw["state_dict"]) == {"module.weight", "module.bias"} and bool((raw["state_dict"]["module.weight"] == 1.0).all())
    m2 = torch.nn.Linear(3, 2)
    fit.ModelCheckpoint.load(str(tmp_path / "best.pt"), m2)
    assert torch.equal(m2.weight, m.weight) and torch.equal(m2.bias, m.bias)


def test_micro_f1_and_k_runs_reduction():
    pred = torch.tensor([[2.0, 1.0], [0.0, 3.0], [1.0, 0.5]])
    assert math.isclose(fit.micro_f1(pred, torch.tensor([0, 1, 1])), 2 / 3, abs_tol=1e-6)
    ml_pred = torch.tensor([[3.0, -3.0], [3.0, 3.0]])             # predicted: [1,0], [1,1]
    ml_true = torch.tensor([[1.0, 0.0], [0.0, 1.0]])              # TP 2, FP 1, FN 0 -> 4 / 5
    assert math.isclose(fit.micro_f1(ml_pred, ml_true, multilabel=True), 0.8)
    out = fit.k_runs(lambda i: dict(final={"Test": 0.5 + 0.1 * i}, best_val_acc=0.6), 3)
    r = out["reduced"]["Test"]
    assert math.isclose(r["mean"], 0.6) and math.isclose(r["std"], (0.02 / 3) ** 0.5) and r["n"] == 3
    assert out["reduced"]["best_val_acc"]["std"] == 0.0
