"""GPU suite: training-trajectory agreement (the accuracy clause's stand-in, SURVEY.md section 8d: no dataset files exist
offline, so 'val accuracy within 0.3 %' cannot be evaluated; what can is that the HIP path and the CPU restatement of the
same training step -- oracle sampler + torch-CPU SAGE + Adam + oracle exp3, oracle/train_ref.py -- walk the same loss
curve from the same initial parameters, batches and sampler stream).

bf16 everywhere (train_lightning.py:596-618): the two sides round their GEMM / SpMM accumulations differently, so
parameters drift apart by bf16 ulps and the curves are compared with a tolerance, not bit for bit.  Two tasks:
  * small features: every EXP3 update factor rounds to 1.0 in bf16 on both sides (exp(x) = 1 for x < ~0.002),
    so the bandit state stays put and the BLOCKS must be identical at every one of the 200 steps;
  * N(0,1) features: the bandit moves; norms that differ by an ulp may move it differently, so blocks are only required to
    be identical at step 0 and the curves to agree after smoothing."""
import pytest
import torch

pytestmark = pytest.mark.gpu

V, E, F, HID, CLASSES, FAN, BATCH, STEPS = 3000, 40000, 24, 32, 4, [64, 32, 16], 32, 200


def _run(cuda, feat_scale):
    import bliss_gnn_amd as bg
    from bliss_gnn_amd.model import SAGE
    from bliss_gnn_amd.synth import chung_lu_csc
    from bliss_gnn_amd.train import TrainStep
    from oracle import bliss_oracle as bo
    from oracle.train_ref import RefTrainStep
    ip, ix, ei = chung_lu_csc(V, E, seed=12)
    gen = torch.Generator().manual_seed(5)
    feats = (torch.randn(V, F, generator=gen) * feat_scale).bfloat16()
    labels = (feats.float() @ torch.randn(F, CLASSES, generator=gen)).argmax(1)           # a learnable task
    batches = [torch.randperm(V, generator=gen)[:BATCH].to(torch.int32) for _ in range(STEPS)]
    torch.manual_seed(0)
    ref = RefTrainStep(bo.CSC(ip, ix, ei), feats, labels, FAN, 0.1, F, HID, CLASSES, lr=0.002, dropout=0.0)
    g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
    g.edata["w"] = bg.normalized_edata(g)
    sampler = bg.PoissonBanditLadiesSampler(FAN, importance_sampling=1, node_embedding="features", eta=0.1, model="sage")
    model = SAGE(F, HID, CLASSES, 3, torch.relu, 0.0).to(cuda).bfloat16()
    model.load_state_dict({k: v.to(cuda) for k, v in ref.model.state_dict().items()})      # same initial parameters
    step = TrainStep(g, sampler, model, lr=0.002)
    ours, theirs, same_blocks = [], [], []
    for i, seeds in enumerate(batches):
        torch.manual_seed(1000 + i)                                                      # the sampler stream of this step
        r_loss, r_blocks = ref(seeds)
        torch.manual_seed(1000 + i)
        loss = step(seeds.to(cuda))
        ours.append(float(loss)); theirs.append(float(r_loss))
        same_blocks.append(all(torch.equal(b.srcdata[bg.NID].cpu().long(), ob.src_nid) and torch.equal(b.src.cpu().long(), ob.src)
                               for b, ob in zip(step.last["mfgs"], r_blocks)))
    sampler.check_errors()
    # (the L1 renormalisation rescales a uniform row to 1/|E|: "moved" = some weight differs from its row's others)
    w = sampler._w_pos.view(torch.int16)
    moved = bool((w != w[:, :1]).any())
    return torch.tensor(ours), torch.tensor(theirs), same_blocks, moved


def _ema(x, a=0.9):
    out, m = [], float(x[0])
    for v in x.tolist():
        m = a * m + (1 - a) * v
        out.append(m)
    return torch.tensor(out)


def test_loss_curve_small_features_identical_blocks(cuda):
    ours, theirs, same, moved = _run(cuda, 0.002)
    assert not moved and all(same)                                  # frozen bandit => the same blocks at all 200 steps
    assert abs(float(ours[0] - theirs[0])) < 0.02                   # identical inputs and parameters at step 0
    assert float((ours - theirs).abs().max()) < 0.08                # bf16 parameter drift only
    assert float(ours[-20:].mean()) <= float(ours[:20].mean())      # and it does not diverge


def test_loss_curve_unit_features_agree(cuda):
    ours, theirs, same, moved = _run(cuda, 1.0)
    assert same[0] and moved                                        # the bandit moves on this task
    assert abs(float(ours[0] - theirs[0])) < 0.05
    assert float((_ema(ours) - _ema(theirs)).abs().max()) < 0.15    # smoothed trajectories coincide
    assert float(ours[-20:].mean()) < 0.8 * float(ours[:20].mean())
    # how long the two sides keep sampling the very same blocks (reported, not required beyond step 0)
    print("identical blocks for the first %d steps, %d of %d overall" % (same.index(False) if False in same else STEPS, sum(same), STEPS))
