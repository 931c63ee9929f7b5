"""The experiment protocol around the hot path: what ``train_lightning.py`` asks of
pytorch_lightning, reduced to plain loops -- Lightning / torchmetrics / tensorboard_reducer are absent on this platform.

  * ``MultiLayerFullNeighborSampler`` / ``NeighborSampler``: the two non-LADIES ``--sampler`` choices
    (train_lightning.py:349-357).  (With them the reference's models fail at ``block.edata["edge_weights"]``,
    model.py:321-329 -- DGL's own samplers attach no such field -- so they are baselines here, with unit weights.)
  * ``fit``: epochs of TrainStep, ``StepLR(gamma=0.01, step_size=5)`` stepped per epoch (:205-216), validation with the same
    sampler (:179-203, :410-422), best-``val_acc`` checkpoint (:620-625), early stop on ``val_acc_target`` / patience
    (:627-634), then the best checkpoint reloaded for the layer-wise full-neighbour inference and the Final Accuracy of the
    three splits (:662-705).  Accuracy = micro-F1 (:68-70).
  * ``k_runs``: mean / std of the final metrics over repeated runs (:711-733).
"""
import copy
import math
import os

import torch

from .bandit_sampler import BlockSampler
from .graph import NID, Block, as_graph
from .ladies_sampler import PoissonLadiesSampler
from .train import BatchLoader, TrainStep, _inputs


# ----------------------------------------------------------------------------------------------- baseline samplers
class MultiLayerFullNeighborSampler(BlockSampler):
    """``dgl.dataloading.MultiLayerFullNeighborSampler(n_layers)`` (train_lightning.py:349-350): every in-edge of every seed,
    layer after layer.  Built on the LADIES kernels with a fanout no candidate set can exceed: the Poisson scale then takes
    its early-out (everything kept, P = 1, ladies_sampler.py:147-148); the edge weights are then set to exactly 1 (the
    Horvitz-Thompson weight bf16(1/deg) * deg is 1 only up to a bf16 rounding) -- the plain mean the reference's inference
    uses (model.py:347-349)."""

    def __init__(self, num_layers):
        super().__init__()
        self.num_layers = int(num_layers)
        self.nodes_per_layer = [-1] * self.num_layers          # (-1 = every in-neighbour, DGL's convention for "no fanout")
        self._inner = None

    def sample_blocks(self, g, seed_nodes, exclude_eids=None):
        g = as_graph(g, self.__dict__.setdefault("_graphs", {}))
        if self._inner is None:
            self._inner = PoissonLadiesSampler([g.num_nodes() + 1] * self.num_layers)
        if "w" not in g.edata:
            from .bandit_sampler import normalized_edata
            g.edata["w"] = normalized_edata(g)
        inp, outp, blocks = self._inner.sample_blocks(g, seed_nodes)
        for b in blocks:
            b.edata["edge_weights"] = torch.ones_like(b.edata["edge_weights"])
        return inp, outp, blocks


class NeighborSampler(BlockSampler):
    """``dgl.dataloading.NeighborSampler(fanouts)`` (train_lightning.py:351-357): up to ``fanout`` in-neighbours per
    destination, uniformly without replacement, per layer.  A baseline outside the hot path: device tensor ops (random key
    per frontier edge, rank inside its column), its own torch generator, no parity claim (DGL draws with its own RNG)."""

    def __init__(self, fanouts, seed=0, **_ignored):
        super().__init__()
        self.fanouts, self.nodes_per_layer = list(fanouts), list(fanouts)
        self._seed, self._gen = seed, None

    def sample_blocks(self, g, seed_nodes, exclude_eids=None):
        g = as_graph(g, self.__dict__.setdefault("_graphs", {}))
        dev = g.device
        if self._gen is None:
            self._gen = torch.Generator(device=dev)
            self._gen.manual_seed(self._seed)
        blocks, seeds = [], seed_nodes.to(torch.int32)
        for fanout in reversed(self.fanouts):
            s64 = seeds.long()
            start, deg = g.indptr[s64], g.indptr[s64 + 1] - g.indptr[s64]
            S, E = s64.numel(), int(deg.sum())
            dst = torch.repeat_interleave(torch.arange(S, device=dev), deg, output_size=E)
            seg = torch.cumsum(deg, 0) - deg
            pos = start[dst] + (torch.arange(E, device=dev) - seg[dst])
            key = dst.double() + torch.rand(E, generator=self._gen, device=dev, dtype=torch.float64) * 0.999999
            order = torch.argsort(key)
            rank = torch.arange(E, device=dev) - seg[dst[order]]
            keep = order[rank < fanout]
            keep = keep[torch.argsort(keep)]                                   # back to column order
            e_dst, e_pos = dst[keep], pos[keep]
            src_g = g.indices[e_pos].long()
            local = torch.full((g.num_nodes(),), -1, dtype=torch.int64, device=dev)
            local[s64] = torch.arange(S, device=dev)
            new = torch.unique(src_g[local[src_g] < 0])
            local[new] = S + torch.arange(new.numel(), device=dev)
            src_nid = torch.cat([s64, new]).to(torch.int32)
            kd = torch.bincount(e_dst, minlength=S)
            indptr = torch.zeros(S + 1, dtype=torch.int32, device=dev)
            indptr[1:] = torch.cumsum(kd, 0)
            eid = g.eid[e_pos] if g.eid is not None else e_pos.to(torch.int32)
            blk = Block(g, src_nid.numel(), S, indptr, local[src_g].to(torch.int32), e_dst.to(torch.int32), e_pos.to(torch.int32), eid, src_nid)
            blk.edata["edge_weights"] = torch.ones(keep.numel(), dtype=torch.bfloat16, device=dev)
            blocks.insert(0, blk)
            seeds = src_nid
        return blocks[0].srcdata[NID], seed_nodes, blocks


def make_sampler(name, fanouts, importance_sampling=1, num_steps=5000, eta=0.1, model="sage"):
    """The sampler-name dispatch of DataModule.__init__ (train_lightning.py:348-370)."""
    from . import BanditLadiesSampler, LadiesSampler, PoissonBanditLadiesSampler, PoissonLadiesSampler as PLS
    if name == "full":
        return MultiLayerFullNeighborSampler(len(fanouts))
    if name == "neighbor":
        return NeighborSampler(fanouts)
    if "ladies" in name and "bandit" not in name:
        return (PLS if "poisson" in name else LadiesSampler)(fanouts)
    if "bandit" in name:
        return (PoissonBanditLadiesSampler if "poisson" in name else BanditLadiesSampler)(
            fanouts, importance_sampling=importance_sampling, node_embedding="features", num_steps=num_steps, eta=eta, model=model)
    raise ValueError("unknown sampler %r" % (name,))


# ----------------------------------------------------------------------------------------------- metrics / control
def micro_f1(pred, labels, multilabel=False):
    """torchmetrics Multiclass / MultilabelF1Score(average='micro') (train_lightning.py:68-70): accuracy for single-label
    predictions; for multilabel 2 TP / (2 TP + FP + FN) over all (node, class) pairs at threshold 0.5."""
    if not multilabel:
        return float((pred.argmax(1) == labels.long()).float().mean())
    hit = torch.sigmoid(pred.float()) > 0.5
    y = labels > 0.5
    tp, fp, fn = float((hit & y).sum()), float((hit & ~y).sum()), float((~hit & y).sum())
    return 2 * tp / max(2 * tp + fp + fn, 1.0)


class StepLR:
    """``th.optim.lr_scheduler.StepLR(optimizer, gamma=0.01, step_size=5)`` stepped once per EPOCH (train_lightning.py:205-216,
    Lightning's default interval): lr x 0.01 every 5 epochs.  Works with any optimiser exposing ``param_groups``."""

    def __init__(self, optimizer, step_size=5, gamma=0.01):
        self.opt, self.step_size, self.gamma, self.epoch = optimizer, int(step_size), float(gamma), 0
        self.base = [g["lr"] for g in optimizer.param_groups]

    def lr_at(self, epoch):
        return [b * self.gamma ** (epoch // self.step_size) for b in self.base]

    def step(self):
        self.epoch += 1
        for g, lr in zip(self.opt.param_groups, self.lr_at(self.epoch)):
            g["lr"] = lr
        if hasattr(self.opt, "sync_lr"):
            self.opt.sync_lr()


class EarlyStopping:
    """``EarlyStopping(monitor='val_acc', stopping_threshold=val_acc_target, mode='max', patience=...)`` (:627-634)."""

    def __init__(self, stopping_threshold=1.0, patience=1000):
        self.threshold, self.patience, self.best, self.bad = stopping_threshold, int(patience), -math.inf, 0

    def should_stop(self, val_acc):
        if val_acc > self.best:
            self.best, self.bad = val_acc, 0
        else:
            self.bad += 1
        # Lightning's stopping_threshold in mode='max' is STRICT: stop once the monitored value is better than the threshold
        return val_acc > self.threshold or self.bad >= self.patience


class ModelCheckpoint:
    """``ModelCheckpoint(monitor='val_acc', save_top_k=1, mode='max')`` (:620-625): keep the best parameters (in memory, and
    on disk when a path is given; the EXP3 state is NOT part of it, as in the reference -- bandit_sampler.py:43).

    On disk the file has the layout of the reference's Lightning ``.ckpt`` as far as its reload path reads it
    (train_lightning.py:64, :671-682: ``load_from_checkpoint`` takes ``checkpoint['state_dict']``, whose keys carry the
    ``module.`` prefix of ``ModelLightning.module``): ``{'state_dict': {'module.<name>': tensor}, 'epoch', 'monitor', 'best'}``
    -- tensors and plain numbers only, so ``torch.load(path, weights_only=True)`` reads it."""

    PREFIX = "module."

    def __init__(self, path=None):
        self.path, self.best, self.state, self.epoch = path, -math.inf, None, -1

    def update(self, val_acc, model, epoch=-1):
        if val_acc > self.best:
            self.best, self.epoch = val_acc, int(epoch)
            self.state = copy.deepcopy({k: v.detach().clone() for k, v in model.state_dict().items()})
            if self.path:
                os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
                torch.save({"state_dict": {self.PREFIX + k: v for k, v in self.state.items()}, "epoch": self.epoch,
                            "monitor": "val_acc", "best": float(self.best)}, self.path)
            return True
        return False

    def restore(self, model):
        if self.state is not None:
            model.load_state_dict(self.state)

    @classmethod
    def load(cls, path, model, strict=True):
        """Load a checkpoint written by ``update`` -- or a Lightning ``.ckpt`` of the reference whose tensors the safe loader
        accepts -- into ``model`` (``strict=False`` like the reference's GCN reload, train_lightning.py:675-680)."""
        ck = torch.load(path, map_location="cpu", weights_only=True)
        sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
        sd = {(k[len(cls.PREFIX):] if k.startswith(cls.PREFIX) else k): v for k, v in sd.items()}
        return model.load_state_dict(sd, strict=strict)


@torch.no_grad()
def evaluate(g, sampler, model, ids, batch_size, multilabel=False, loss_fn=None):
    """validation_step over a split (train_lightning.py:179-203): the same sampler object, no bandit update, no optimiser."""
    was = model.training
    model.eval()
    preds, labels, losses = [], [], []
    for seeds in BatchLoader(ids, batch_size, shuffle=False, drop_last=False):
        _, _, mfgs = sampler.sample(g, seeds)
        pred = model(mfgs, _inputs(model, mfgs))
        y = mfgs[-1].dstdata["labels"]
        preds.append(pred.float()); labels.append(y)
        if loss_fn is not None:
            losses.append(float(loss_fn(pred, y)) * seeds.numel())
    model.train(was)
    pred, y = torch.cat(preds), torch.cat(labels)
    return micro_f1(pred, y, multilabel), (sum(losses) / max(ids.numel(), 1) if losses else None)


def fit(g, sampler, model, train_nid, val_nid, test_nid=None, batch_size=1024, lr=0.002, max_epochs=10, max_steps=None,
        multilabel=False, val_acc_target=1.0, early_stopping_patience=1000, checkpoint_path=None, seed=0, log=None):
    """One run of ``trainer.fit`` + the final evaluation (train_lightning.py:640-705).  Returns a dict of metrics."""
    g = as_graph(g)
    step = TrainStep(g, sampler, model, lr=lr, multilabel=multilabel)
    sched, stopper, ckpt = StepLR(step.opt, 5, 0.01), EarlyStopping(val_acc_target, early_stopping_patience), ModelCheckpoint(checkpoint_path)
    loader = BatchLoader(train_nid, batch_size, shuffle=True, drop_last=True, seed=seed)
    history, n_steps = [], 0
    for epoch in range(max_epochs):
        model.train()
        tot, cnt = 0.0, 0
        for seeds in loader:
            tot += float(step(seeds)); cnt += 1; n_steps += 1
            if max_steps is not None and n_steps >= max_steps:
                break
        val_acc, val_loss = evaluate(g, sampler, model, val_nid, batch_size, multilabel, step.loss_fn)
        ckpt.update(val_acc, model, epoch)
        history.append(dict(epoch=epoch, train_loss=tot / max(cnt, 1), val_acc=val_acc, val_loss=val_loss, lr=step.opt.param_groups[0]["lr"]))
        if log:
            log(history[-1])
        sched.step()                                                             # per epoch (:205-216)
        if stopper.should_stop(val_acc) or (max_steps is not None and n_steps >= max_steps):
            break
    ckpt.restore(model)                                                          # the best val_acc checkpoint (:662-685)
    final = {}
    if hasattr(model, "inference") and "features" in g.ndata:
        pred = model.inference(g)                                                # :686-693 layer-wise full-neighbour inference
        for name, nid in (("Train", train_nid), ("Validation", val_nid), ("Test", test_nid)):
            if nid is not None and nid.numel():
                final[name] = micro_f1(pred[nid.long()].float(), g.ndata["labels"][nid.long()], multilabel)    # :694-705
    return dict(history=history, best_val_acc=ckpt.best, steps=n_steps, final=final)


def k_runs(run_fn, k):
    """``--k-runs`` (train_lightning.py:565, :711-733): repeat a run, reduce every final metric to mean / std (what
    tensorboard_reducer writes with reduce_ops = ('mean', 'std'); population std, as numpy's default)."""
    outs = [run_fn(i) for i in range(k)]
    keys = sorted({("final", n) for o in outs for n in o["final"]} | {("best_val_acc", None)})
    red = {}
    for kind, n in keys:
        xs = [o["final"][n] if kind == "final" else o["best_val_acc"] for o in outs if kind != "final" or n in o["final"]]
        m = sum(xs) / len(xs)
        red[n or "best_val_acc"] = dict(mean=m, std=(sum((x - m) ** 2 for x in xs) / len(xs)) ** 0.5, n=len(xs))
    return dict(runs=outs, reduced=red)
