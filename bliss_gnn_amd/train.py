"""A lean training loop that reproduces the reference's call sites around the hot path.

pytorch_lightning / dgl.dataloading.DataLoader are not available on this platform, so this module
replays, in order, exactly what they do to the sampler and the model each step
(train_lightning.py:100-168 training_step, :205-216 optimiser, :396-408 loader, :463-471 callback):

    seeds  = next batch of train ids (shuffled per epoch, drop_last)           DataLoader
    input_nodes, output_nodes, mfgs = sampler.sample(g, seeds)                 BlockSampler.sample
    x = mfgs[0].srcdata['features'];  y = mfgs[-1].dstdata['labels']          training_step :138-139
    loss = loss_fn(model(mfgs, x), y);  loss.backward();  optimiser.step()     :141-142 + Lightning
    sampler.exp3(mfgs, g)                                                      BatchSizeCallback :469-471
"""
import contextlib
import os
import time

import torch
import torch.nn as nn

from . import _lib


class BatchLoader:
    """dgl.dataloading.DataLoader(g, train_nid, sampler, batch_size, shuffle=True, drop_last=True)
    reduced to its id stream (train_lightning.py:396-408).  Shuffles on the ids' device with its own
    generator so that the sampler's CPU random stream is untouched."""

    def __init__(self, ids, batch_size, shuffle=True, drop_last=True, seed=2):
        self.ids, self.bs, self.shuffle, self.drop_last = ids, int(batch_size), shuffle, drop_last
        self.gen = torch.Generator(device=ids.device)
        self.gen.manual_seed(seed)

    def __len__(self):
        n = self.ids.numel()
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        ids = self.ids
        if self.shuffle:
            ids = ids[torch.randperm(ids.numel(), generator=self.gen, device=ids.device)]
        for i in range(len(self)):
            yield ids[i * self.bs:(i + 1) * self.bs]

    def forever(self):
        while True:
            yield from iter(self)


def _cap_kw():
    """Keyword arguments of every ``torch.cuda.graph`` capture here: with a process group alive, ProcessGroupNCCL's watchdog thread
    queries its events at any time, and a capture in the default "global" error mode turns such a query from ANOTHER thread into
    "operation not permitted when stream is capturing" (seen once captures became frequent: bench.py --dist shards aborted)."""
    try:
        import torch.distributed as _d
        return {"capture_error_mode": "thread_local"} if _d.is_available() and _d.is_initialized() else {}
    except Exception:                                            # noqa: BLE001
        return {}


def _inputs(model, mfgs):
    """``mfgs[0].srcdata['features']`` (train_lightning.py:138); for a model whose first layer gathers the rows as its operand
    load (model.SAGE on the MFMA path) the not-yet-gathered (table, ids) pair instead."""
    if getattr(model, "accepts_lazy_rows", False):
        return mfgs[0].srcdata.lazy("features")
    return mfgs[0].srcdata["features"]


def _ce_loss():
    if os.environ.get("BLISS_FUSED_CE", "1") == "0":
        return nn.CrossEntropyLoss()
    from .nn import CrossEntropyLoss
    return CrossEntropyLoss()


def _loss_backward(loss_fn, pred, labels, opt):
    """loss = loss_fn(pred, labels); opt.zero_grad(); loss.backward() (train_lightning.py:142 + Lightning).  The one-launch
    cross-entropy hands d loss / d pred over with the loss, so its route skips the loss node.  Returns the detached loss."""
    opt.zero_grad(set_to_none=True)
    if hasattr(loss_fn, "backward_from"):
        return loss_fn.backward_from(pred, labels)
    loss = loss_fn(pred, labels)
    loss.backward()
    return loss.detach()


def make_adam(model, lr, capturable=False):
    """th.optim.Adam(self.parameters(), lr) (train_lightning.py:206).  For the reference's precision (bf16 module on the GPU,
    :596-618) this is the one-launch gfx950 Adam of csrc/optim.hip; anything else gets torch's own."""
    ps = list(model.parameters())
    if ps and all(p.is_cuda and p.dtype == torch.bfloat16 and p.is_contiguous() for p in ps) and len(ps) <= _lib.ADAM_MAX_TENSORS:
        from .optim import Adam
        return Adam(ps, lr=lr)
    if capturable:
        return torch.optim.Adam(ps, lr=lr, capturable=True, fused=True)
    return torch.optim.Adam(ps, lr=lr)


class TrainStep:
    """One optimiser step of ModelLightning (train_lightning.py:50-216) with the bandit callback."""

    def __init__(self, g, sampler, model, lr=0.002, multilabel=False, bandit=True, grad_sync=None, exp3_sync=None):
        self.g, self.sampler, self.model = g, sampler, model
        self.loss_fn = nn.BCEWithLogitsLoss() if multilabel else _ce_loss()             # :77-79
        self.opt = make_adam(model, lr)                                                  # :206
        self.bandit = bandit and hasattr(sampler, "exp3")          # train_lightning.py:469: only for the bandit samplers
        self.grad_sync, self.exp3_sync = grad_sync, exp3_sync
        self.num_steps = 0
        self.w = 0.99                                                                    # :76
        n_layers = len(sampler.nodes_per_layer)
        self.cum_sampled_nodes = [0.0] * (n_layers + 1)
        self.cum_sampled_edges = [0.0] * n_layers
        self.last = {}

    def _ema(self, mfgs):
        self.num_steps += 1                                                              # :103
        for i, mfg in enumerate(mfgs):                                                   # :104-110
            self.cum_sampled_nodes[i] = self.cum_sampled_nodes[i] * self.w + mfg.num_src_nodes()
            self.cum_sampled_edges[i] = self.cum_sampled_edges[i] * self.w + mfg.num_edges()
        i = len(mfgs)
        self.cum_sampled_nodes[i] = self.cum_sampled_nodes[i] * self.w + mfgs[-1].num_dst_nodes()   # :127-129

    def num_sampled_edges(self, i):                                                      # :91-98
        return self.cum_sampled_edges[i] * (1 - self.w) / (1 - self.w ** self.num_steps)

    def num_sampled_nodes(self, i):                                                      # :82-89
        return self.cum_sampled_nodes[i] * (1 - self.w) / (1 - self.w ** self.num_steps)

    def __call__(self, seeds):
        input_nodes, output_nodes, mfgs = self.sampler.sample(self.g, seeds)
        self._ema(mfgs)
        batch_inputs = _inputs(self.model, mfgs)                                         # :138
        batch_labels = mfgs[-1].dstdata["labels"]                                        # :139
        batch_pred = self.model(mfgs, batch_inputs)                                      # :141
        loss = _loss_backward(self.loss_fn, batch_pred, batch_labels, self.opt)          # :142
        if self.grad_sync is not None:
            self.grad_sync(self.model)
        self.opt.step()
        if self.bandit:
            if self.exp3_sync is not None:
                self.exp3_sync(self.sampler, mfgs, self.g)
            else:
                self.sampler.exp3(mfgs, self.g)                                          # :469-471
        self.last = dict(loss=loss, mfgs=mfgs, pred=batch_pred, labels=batch_labels)
        return loss


def _enable_gemm_tuning():
    """Let PyTorch's TunableOp measure the rocBLAS / hipBLASLt solutions of every GEMM shape it meets from now on.  Returns
    False (and leaves the library defaults in place) if this PyTorch build cannot."""
    try:
        tn = torch.cuda.tunable
        tn.enable(True)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(30)
        tn.set_max_tuning_iterations(20)
        tn.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "bliss_tunableop_%d.csv" % os.getpid()))
        return True
    except Exception as e:                                   # noqa: BLE001 -- tuning is an optimisation, never a requirement
        import warnings
        warnings.warn("GEMM tuning unavailable (%r); using the library defaults" % (e,))
        return False


class GraphedTrainStep:
    """The same step as TrainStep, recorded ONCE into a HIP graph and replayed: sampler kernels, feature gather,
    SAGE forward / backward, Adam and the EXP3 update with no host work in between.

    The reference launches ~350 kernels and syncs >= 24 times per step from Python; the
    eager TrainStep above still pays ~150 launches and one mid-step sync.  Replay needs static shapes, so the
    blocks are padded to capacities learned from a few eager steps (``calibrate``); true sizes stay on the device
    and come back with the step's single end-of-step sync.  Results are bit-identical to the eager path."""

    def __init__(self, g, sampler, model, batch_size, lr=0.002, multilabel=False, distributed=False):
        self.g, self.sampler, self.model, self.bs = g, sampler, model, int(batch_size)
        self.distributed = distributed          # replicas: gradient all-reduce + EXP3 exchange recorded in the graph too
        self.loss_fn = nn.BCEWithLogitsLoss() if multilabel else _ce_loss()
        # ONE launch for all parameter tensors (csrc/optim.hip; torch's foreach path is ~40 launches of >= 5 us inside a graph,
        # its fused multi-tensor kernel ~50 us)
        self.opt = make_adam(model, lr, capturable=True)
        self.seeds = torch.zeros(self.bs, dtype=torch.int32, device=g.device)
        self.graph = None
        self.num_steps = 0
        self.last_counts = None
        self.loss = None

    def calibrate(self, loader, steps=8, k_margin=1.5, b_margin=3.0):
        """Run eager sampling to learn per-layer sizes, then fix the static capacities."""
        L = len(self.sampler.nodes_per_layer)
        mx = [dict(K=0, B=0, E=0) for _ in range(L)]
        for _ in range(steps):
            _, _, blocks = self.sampler.sample_blocks(self.g, next(loader))
            for n, b in enumerate(reversed(blocks)):                      # sampling order
                mx[n]["K"] = max(mx[n]["K"], b.num_src_nodes())
                mx[n]["B"] = max(mx[n]["B"], b.num_edges())
                mx[n]["E"] = max(mx[n]["E"], b._counts.E)
        if self.distributed:                       # the exchanged lists are capacity-sized: every rank needs the same capacities
            import torch.distributed as dist
            t = torch.tensor([[m["K"], m["B"], m["E"]] for m in mx], dtype=torch.int64, device=self.g.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mx = [dict(K=int(k), B=int(b), E=int(e)) for k, b, e in t.tolist()]
        fan = [self.sampler.nodes_per_layer[b] for b in reversed(range(L))]
        self._margins, self._hw = (k_margin, b_margin), [dict(m) for m in mx]
        self.sampler._engine.set_static_caps(self.bs, fan, mx, k_margin, b_margin)

    def _body(self):
        input_nodes, output_nodes, mfgs = self.sampler.sample_blocks_static(self.g, self.seeds)
        x = _inputs(self.model, mfgs)
        y = mfgs[-1].dstdata["labels"]
        pred = self.model(mfgs, x)
        loss = _loss_backward(self.loss_fn, pred, y, self.opt)
        bandit = hasattr(self.sampler, "exp3")                     # train_lightning.py:469: only for the bandit samplers
        if self.distributed:
            from . import dist as bdist
            bdist.allreduce_gradients(self.model)
            self.opt.step()
            if bandit:
                bdist.exp3_all_ranks_static(self.sampler, mfgs, self.g)
        else:
            self.opt.step()
            if bandit:
                self.sampler.exp3(mfgs, self.g)
        # detach: a live autograd graph would pin the warm-up stream's AccumulateGrad nodes into the capture
        return loss.detach()

    def _finish(self):
        torch.cuda.current_stream().synchronize()
        self.last_counts = self.sampler.finish_static()
        self.num_steps += 1

    def capture(self, loader, warmup=3, tune_gemm=False):
        """Eager static-shape warm-up steps on a side stream (allocator + autograd warm), then capture.

        ``tune_gemm``: let PyTorch's TunableOp pick the rocBLAS / hipBLASLt solution for every dense transform during
        the warm-up (the shapes are static, so each is tuned once); the tall-skinny weight-gradient GEMMs otherwise get
        a default tile that fills only a fraction of the 256 CUs."""
        eng = self.sampler._engine
        tune_gemm = tune_gemm and _enable_gemm_tuning()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.seeds.copy_(next(loader))
                eng.stage_rng_from_torch()
                self.loss = self._body()
                self._finish()
        torch.cuda.current_stream().wait_stream(side)
        if tune_gemm:
            torch.cuda.tunable.tuning_enable(False)        # keep using the tuned solutions, stop measuring
        self.loss = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        self.seeds.copy_(next(loader))
        eng.stage_rng_from_torch()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **_cap_kw()):
            self.loss = self._body()
        # the capture itself executed nothing: replay once so that this batch is a real step
        self.graph.replay()
        self._finish()

    def __call__(self, seeds):
        self.seeds.copy_(seeds)
        self.sampler._engine.stage_rng_from_torch()
        self.graph.replay()
        self._finish()
        return self.loss

    def eager_step(self, seeds):
        """The same static-shape step launched kernel by kernel (used to time individual kernels)."""
        self.seeds.copy_(seeds)
        self.sampler._engine.stage_rng_from_torch()
        loss = self._body()
        self._finish()
        return loss

    def sizes(self):
        """Per block (input-most first): the true S, E, C, K, B of the last step."""
        return [dict(S=c.S, E=c.E, C=c.C, K=c.K, B=c.B) for c in reversed(self.last_counts)]

    def _graph_attrs(self):
        return ("graph",)

    def close(self):
        """Quiesce the device and destroy the captured graphs NOW.  A distributed run must call this before
        ``destroy_process_group()``: the graphs hold RCCL nodes, and graphs that outlive their communicator (destroyed at
        interpreter exit, after the process group) abort the process."""
        import gc
        torch.cuda.synchronize()
        for name in self._graph_attrs():
            v = getattr(self, name, None)
            if isinstance(v, list):
                setattr(self, name, [None] * len(v))
            elif v is not None:
                setattr(self, name, None)
        self.loss = None
        if hasattr(self, "losses"):
            self.losses = None
        gc.collect()
        torch.cuda.synchronize()


class PipelinedTrainStep(GraphedTrainStep):
    """Two train steps per call, software-pipelined: while the backward pass and Adam of batch ``a`` run on one stream,
    the sampler already builds the blocks of batch ``b`` on another (and vice versa); every piece is a replayed HIP graph.

        F(a) X(a) [ S(b) || B(a) ]  F(b) X(b) [ S(a') || B(b) ]          F forward+loss, X exp3 update, B backward+Adam, S sample

    The sampler is a long chain of small latency- and atomic-bound kernels that leaves most of the chip idle; the
    dense backward fills that idle capacity.  Nothing is reordered that depends on anything else: S(b) needs the EXP3
    weights after X(a) (it comes after it) and not the parameters; X reads only what the forward left on the blocks
    (embed_norm, q_ij), so running it before B changes no value -- every step computes exactly what the sequential
    loop computes, bit for bit, and torch's CPU generator is consumed in the same order (S(b) then S(a')).

    How the pieces are ordered (``use_flags``, the default): every graph launch costs ~20 us on the stream it is launched
    on, and an event between two kernels cuts a graph in two.  So the critical chain F X S of one step is ONE graph on the
    main stream, and the work beside it is handed off through device flags instead of events (bliss_flag_wait): the
    backward graph (second stream) starts with a wait for the flag the sampler's first kernel raises (= X has finished);
    the blocks of all but the last-sampled layer are a third graph on a third stream, each behind a wait for the flag the
    NEXT layer's first kernel raises (= this layer's draw has finished) -- block n shares no scratch with the later layers
    (one scratch set per layer in the engine), so the critical path loses those block passes.  Inside ONE graph parallel
    branches would share a hardware queue on ROCm 7.2, hence three graphs on three real streams.  ``capture`` checks that
    the flags arrive (streams that happen to share a hardware queue would make a waiting kernel block its producer) and
    otherwise falls back to event ordering: F+X, S and B as separate graphs (BLISS_PIPELINE_FLAGS=0 forces that).

    One call = two optimiser steps on two batches; the batch sampled last is trained by the next call (``drain``
    trains the final one)."""

    def __init__(self, g, sampler, model, batch_size, lr=0.002, multilabel=False, distributed=False):
        super().__init__(g, sampler, model, batch_size, lr, multilabel, distributed)
        self.seeds2 = [torch.zeros(self.bs, dtype=torch.int32, device=g.device) for _ in range(2)]
        self.mfgs = [None, None]
        # backward pass + Adam.  Measured and dropped: a high-priority stream (round 2: no different), a LOWEST-priority stream made
        # through the HIP runtime (round 3: 1490.1 vs 1490.0 steps/s) and a CU-masked stream (hipExtStreamCreateWithCUMask, so that
        # the backward pass leaves CUs to the sampler's latency-bound chain: 2.09 ms per step even with the full mask)
        self.side = torch.cuda.Stream()
        self.third = torch.cuda.Stream()         # blocks of all but the last-sampled layer (flag mode)
        self._fwd_done, self._bwd_done, self._blk_done = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self._seed_ev = torch.cuda.Event()
        self.use_flags = os.environ.get("BLISS_PIPELINE_FLAGS", "1") != "0"
        self.losses = None
        self.last_counts2 = None
        # BLISS_NORM_DEFER=1: F.normalize's pass over the bandit rows beside the next forward pass instead of in front of the
        # next sampler (bandit_sampler.normalize_pending; flag mode only, switched on by capture()).  Off by default: same bits,
        # but measured no faster on the Reddit-like loop once the pass itself took 85 us instead of 129 (0.820 / 0.824 against
        # 0.823 / 0.836 ms per step, run-to-run noise +-0.008) -- beside the forward pass it still competes for HBM, and the
        # hand-off adds two small kernels to every step.  Replicas always keep the immediate pass: their update lists are applied
        # after an exchange, outside the fused exp3 step.
        self._defer_wanted = (not distributed and hasattr(sampler, "normalize_pending")
                              and os.environ.get("BLISS_NORM_DEFER", "0") != "0")
        self._defer = False
        self.g_norm = None
        self._flag_boundary = os.environ.get("BLISS_FLAG_BOUNDARY", "1") != "0"
        # The input layer's block (the LAST one the sampler builds: ~90 us at the end of its chain) built on the third stream
        # beside the next step's first transform, which needs the kept-node list only; the first aggregation waits for a
        # flag (nn._wait_block).  SAGE only: its first reader of the block's arrays is that aggregation.
        self._defer_block0 = False
        self.g_blk0 = [None, None]
        self._blk0_done = torch.cuda.Event()

    def _sample(self, slot, chain, external_rng=False, part=None, last_block=True, ready_flag=0):
        return self.sampler.sample_blocks_static(self.g, self.seeds2[slot], slot=slot, chain_rng=chain, external_rng=external_rng,
                                                 part=part, last_block=last_block, ready_flag=ready_flag)[2]

    def _split_forward(self):
        # BLISS_SPLIT_FORWARD=0 keeps the whole forward pass ahead of the bandit update (the round-1 order)
        return (hasattr(self.model, "forward_hidden") and len(getattr(self.model, "layers", ())) > 1 and hasattr(self.sampler, "exp3")
                and os.environ.get("BLISS_SPLIT_FORWARD", "1") != "0")

    FLAG_BLOCK0, FLAG_B_DONE, FLAG_X_DONE = 10, 11, 12    # engine.flags slots (0..L: the sampler's layers; 14: the probe)

    def _forward(self, mfgs, flagged=False):
        """The part of the step the NEXT batch's sampler waits for: the forward pass up to the output layer's input (every
        block's row norms exist from there on, train_lightning.py:232-238 reads nothing else) and the bandit update.  Returns
        what _backward needs to finish the step."""
        if self._defer and not flagged:                            # eager: the rows the previous update left pending, inline
            self.sampler.normalize_pending()
        pending = self._forward_model(mfgs)
        done_flag = None
        if self._defer and flagged:
            # the pass of the PREVIOUS update ran on the third stream (g_norm) and is complete: this graph only starts after
            # that stream's event.  This update tells g_norm when the rows are ready for the next pass.
            self.sampler._pend_maybe = False
            done_flag = self.sampler._engine.flags.data_ptr() + 4 * self.FLAG_X_DONE
        if not hasattr(self.sampler, "exp3"):                      # LADIES samplers keep no bandit state
            return pending
        if self.distributed:
            from . import dist as bdist
            bdist.exp3_all_ranks_static(self.sampler, mfgs, self.g)
        elif done_flag is not None:
            self.sampler.exp3(mfgs, self.g, done_flag=done_flag)
        else:
            self.sampler.exp3(mfgs, self.g)
        return pending

    def _forward_model(self, mfgs):
        if self._split_forward():
            pending = ("hidden", self.model.forward_hidden(mfgs, _inputs(self.model, mfgs)), mfgs)
        else:
            pred = self.model(mfgs, _inputs(self.model, mfgs))
            pending = ("pred", pred, mfgs)
        return pending

    def _backward(self, pending):
        """Output layer + loss (when _forward left them), backward, optimizer.  Returns the detached loss."""
        kind, val, mfgs = pending
        loss = None
        if kind == "hidden" and hasattr(self.model, "forward_last_parts") and hasattr(self.loss_fn, "backward_from_parts"):
            # output layer's sum and the label gather inside the loss kernel (two small launches less on the backward stream)
            lab = mfgs[-1].dstdata
            table = lab._parent["labels"] if getattr(lab, "_parent", None) is not None and "labels" in lab._parent else None
            parts = self.model.forward_last_parts(mfgs, val) if table is not None and not dict.__contains__(lab, "labels") else None
            if parts is not None:
                self.opt.zero_grad(set_to_none=True)
                loss = self.loss_fn.backward_from_parts(parts[0], parts[1], table, lab._index_fn())
        if loss is None:
            pred = self.model.forward_last(mfgs, val) if kind == "hidden" else val
            loss = _loss_backward(self.loss_fn, pred, mfgs[-1].dstdata["labels"], self.opt)
        if self.distributed:
            from . import dist as bdist
            bdist.allreduce_gradients(self.model)
        self.opt.step()
        return loss

    def _pair(self):
        # Eager version (warm-up, kernel-by-kernel timing).  The sampler stays on the origin stream (its random-number
        # generator forks from there); the model runs on the second stream, forward AND backward (autograd replays a node on
        # the stream of its forward).
        main, side = torch.cuda.current_stream(), self.side
        side.wait_stream(main)
        losses = []
        for cur, nxt, chain in ((0, 1, False), (1, 0, True)):
            with torch.cuda.stream(side):
                pending = self._forward(self.mfgs[cur])      # F + X
            main.wait_stream(side)                           # the sampler needs the EXP3 weights X just wrote
            self.mfgs[nxt] = self._sample(nxt, chain)        # S, beside ...
            with torch.cuda.stream(side):
                losses.append(self._backward(pending))       # ... B
                side.wait_stream(main)                       # the next forward needs the blocks S built
        main.wait_stream(side)
        return tuple(losses)

    def prime(self, seeds):
        """Sample the first batch (slot 0) so that the pipeline has something to train on."""
        self.seeds2[0].copy_(seeds)
        self.sampler._engine.stage_rng_from_torch()
        self.mfgs[0] = self._sample(0, False)
        torch.cuda.current_stream().synchronize()
        self.sampler.finish_static(0, commit=True)

    def _finish_pair(self, check_flags=True):
        torch.cuda.current_stream().synchronize()
        if self._defer:                                  # whoever looks at the rows between two calls finds them in _w_pos
            self.sampler._pend_maybe = True
            self.sampler._settle()
            torch.cuda.current_stream().synchronize()
        c1 = self.sampler.finish_static(1, commit=False)
        c0 = self.sampler.finish_static(0, commit=True)
        self.last_counts2 = [c1, c0]                     # the two batches sampled by this replay, in sampling order
        self.last_counts = c0
        self.num_steps += 2
        if check_flags and self.graph and self.use_flags and int(self.sampler._engine.flag_err.item()):
            raise RuntimeError("a cross-stream flag never arrived (bliss_flag_wait timed out): the pipelined results are invalid")

    def _load(self, loader):
        self.seeds2[1].copy_(next(loader))               # S(b) runs first, then S(a')
        self.seeds2[0].copy_(next(loader))
        self.sampler._engine.stage_rng_from_torch()

    def capture(self, loader, warmup=2, tune_gemm=False):
        eng = self.sampler._bind(self.g)
        L = len(self.sampler.nodes_per_layer)
        if torch.cuda.current_stream() != torch.cuda.default_stream():
            import warnings
            warnings.warn("PipelinedTrainStep: captured from a non-default stream; HIP maps streams onto a few hardware queues and the "
                          "critical chain then shares one with a side stream (measured 1.08 instead of 0.70 ms per step)")
        if self.use_flags and not self._flags_usable():       # (before the warm-up: autograd remembers the streams it ran on)
            import warnings
            warnings.warn("PipelinedTrainStep: streams do not run side by side here (a profiler serialising kernels?); "
                          "ordering the graphs with events instead of device flags")
            self.use_flags = False
        if self.use_flags:
            eng.scratch_sets = max(eng.scratch_sets, L)      # block n then shares no scratch with any later layer
        self._defer = self._defer_wanted and self.use_flags
        if self._defer:
            self.sampler.enable_deferred_normalize(True)
        self._defer_block0 = (self.use_flags and self._flag_boundary and L > 1 and L < self.FLAG_BLOCK0 and type(self.model).__name__ == "SAGE"
                              and self._split_forward() and os.environ.get("BLISS_DEFER_BLOCK0", "1") != "0")
        tune_gemm = tune_gemm and _enable_gemm_tuning()
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            self.prime(next(loader))
            for _ in range(warmup):
                self._load(loader)
                self.losses = self._pair()
                self._finish_pair()
        torch.cuda.current_stream().wait_stream(warm)
        if tune_gemm:
            torch.cuda.tunable.tuning_enable(False)
        self.losses = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        self._capture_graphs(loader)
        if self.use_flags and int(eng.flag_err.item()):
            import warnings
            warnings.warn("PipelinedTrainStep: a cross-stream flag timed out (streams sharing a hardware queue?); "
                          "falling back to event-ordered graphs")
            eng.flag_err.zero_()
            eng.flags.zero_()
            self.use_flags = False
            self._capture_graphs(loader)

    def _probe(self, st, flag_index):
        """One wait on ``st`` enqueued FIRST, the raise on the main stream afterwards: completes without a timeout only if
        the two streams really run side by side."""
        eng = self.sampler._engine
        main = torch.cuda.current_stream()
        torch.cuda.synchronize()
        eng.flag_err.zero_()
        eng.flags.zero_()
        torch.cuda.synchronize()
        f = eng.flags.data_ptr() + 4 * flag_index
        _lib.check(_lib.lib.bliss_flag_wait(f, eng.flag_err.data_ptr(), st.cuda_stream), "bliss_flag_wait")
        _lib.check(_lib.lib.bliss_flag_raise(f, main.cuda_stream), "bliss_flag_raise")
        torch.cuda.synchronize()
        ok = int(eng.flag_err.item()) == 0
        eng.flag_err.zero_()
        eng.flags.zero_()
        torch.cuda.synchronize()
        return ok

    def _flags_usable(self, tries=4):
        """Probe with harmless kernels before relying on device flags for ordering: a flag that times out in the real loop
        would let a consumer run before its producer.  HIP multiplexes streams onto a few hardware queues (4 by default);
        a stream that shares the main stream's queue cannot wait for it, so such a stream is replaced by the next one of
        PyTorch's pool and probed again.  Under a profiler that serialises kernels (rocprofv3 --pmc) no stream passes."""
        for name in ("side", "third"):
            for _ in range(tries):
                if self._probe(getattr(self, name), 14):
                    break
                setattr(self, name, torch.cuda.Stream())
            else:
                return False
        return True

    def _capture_graphs(self, loader):
        # Several graphs, not one: a HIP graph with the sampler and the backward pass as parallel branches is executed with
        # both branches on one hardware queue (ROCm 7.2), i.e. not overlapped.  Replaying them from different real streams
        # gives the overlap (see _half).
        eng = self.sampler._engine
        L = len(self.sampler.nodes_per_layer)
        self._load(loader)
        side = self.side
        pool = torch.cuda.graph_pool_handle()
        self.graph = None
        self.g_main, self.g_fwd, self.g_bwd, self.g_smp, self.g_blk, self.g_blk0 = ([None, None] for _ in range(6))
        self.g_norm = None
        held, out = [None, None], [None, None]
        st_ = lambda: torch.cuda.current_stream().cuda_stream
        if self._defer:
            # F.normalize's pass, for the third stream: as soon as the bandit update has decided which rows need it, write them
            # renormalised into their other buffers -- beside the sampler, which keeps reading the old ones (dividing on the fly)
            # until the pass has switched a row over
            self.g_norm = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_norm, pool=pool, **_cap_kw()):
                _lib.check(_lib.lib.bliss_flag_wait(eng.flags.data_ptr() + 4 * self.FLAG_X_DONE, eng.flag_err.data_ptr(), st_()), "bliss_flag_wait")
                self.sampler.normalize_pending()
        for cur, nxt, chain in ((0, 1, False), (1, 0, True)):
            # (the sampler is recorded without its generator: that one is launched ahead of time, see _replay / run)
            self.g_bwd[cur] = torch.cuda.CUDAGraph()
            if self.use_flags:
                self.g_main[cur] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_main[cur], pool=pool, stream=side, **_cap_kw()):
                    if self._flag_boundary:
                        # "the previous step's backward pass, Adam and early blocks are done", as a device flag: the graph is
                        # launched ahead and its first kernel waits ~3 us past the raise; a stream-event wait in front of the
                        # graph launch cost ~30 us from the end of B to the first kernel of F
                        _lib.check(_lib.lib.bliss_flag_wait(eng.flags.data_ptr() + 4 * self.FLAG_B_DONE, eng.flag_err.data_ptr(), st_()),
                                   "bliss_flag_wait")
                    if self._defer_block0:              # (the wait itself is recorded by the first aggregation over that block)
                        self.mfgs[cur][0]._ready = (eng.flags.data_ptr() + 4 * self.FLAG_BLOCK0, eng.flag_err.data_ptr())
                    held[cur] = self._forward(self.mfgs[cur], flagged=True)                     # F + X
                    self.mfgs[cur][0]._ready = None
                    self.mfgs[nxt] = self._sample(nxt, chain, external_rng=True, part="main",   # S without the early blocks
                                                  last_block=not self._defer_block0)
                if L > 1:
                    self.g_blk[nxt] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g_blk[nxt], **_cap_kw()):
                        self._sample(nxt, chain, external_rng=True, part="early_blocks")
                if self._defer_block0:
                    self.g_blk0[nxt] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g_blk0[nxt], **_cap_kw()):
                        # (raises FLAG_BLOCK0 itself, before it sorts the by-source index the backward pass will read)
                        early = os.environ.get("BLISS_BLOCK0_EARLY_FLAG", "1") != "0"
                        self._sample(nxt, chain, external_rng=True, part="last_block",
                                     ready_flag=eng.flags.data_ptr() + 4 * self.FLAG_BLOCK0 if early else 0)
                        if not early:
                            _lib.check(_lib.lib.bliss_flag_raise(eng.flags.data_ptr() + 4 * self.FLAG_BLOCK0, st_()), "bliss_flag_raise")
                with torch.cuda.graph(self.g_bwd[cur], pool=pool, stream=side, **_cap_kw()):
                    # B may start once S has (flag 0 is raised by the sampler's first kernel: F and X have completed)
                    _lib.check(_lib.lib.bliss_flag_wait(eng.flags.data_ptr(), eng.flag_err.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream), "bliss_flag_wait")
                    out[cur] = self._backward(held[cur])
            else:
                self.g_fwd[cur], self.g_smp[nxt] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_fwd[cur], pool=pool, stream=side, **_cap_kw()):
                    held[cur] = self._forward(self.mfgs[cur])
                with torch.cuda.graph(self.g_smp[nxt], **_cap_kw()):
                    self.mfgs[nxt] = self._sample(nxt, chain, external_rng=True)
                with torch.cuda.graph(self.g_bwd[cur], pool=pool, stream=side, **_cap_kw()):
                    out[cur] = self._backward(held[cur])
        self.losses = tuple(out)
        self.graph = True
        self._replay()                                   # the captures themselves executed nothing
        self._finish_pair(check_flags=False)             # (capture() looks at the flags itself)

    def _half(self, cur, nxt, on_side=None):
        """Enqueue F(cur) X(cur) [ S(nxt) || B(cur) ].  The generator of S(nxt) has been started by the caller.
        ``on_side``: extra work for the backward pass's stream, after B."""
        main, side = torch.cuda.current_stream(), self.side
        if not (self.use_flags and self._flag_boundary):
            main.wait_event(self._bwd_done)              # parameters after the previous step's Adam
        if self.use_flags:
            if self._defer_block0:
                # B(cur) reads the by-source index of its input block, sorted behind that block's ready flag in the PREVIOUS
                # half's g_blk0 (this wait must precede the record further down, which is this half's)
                side.wait_event(self._blk0_done)
            self.g_main[cur].replay()                    # F + X + S: one graph on the critical stream
            if self.g_norm is not None or self.g_blk[nxt] is not None:
                with torch.cuda.stream(self.third):
                    if self.g_norm is not None:
                        self.g_norm.replay()             # F.normalize's pass: waits for X, runs beside S
                    if self.g_blk[nxt] is not None:
                        self.g_blk[nxt].replay()         # early blocks of S: each waits for the flag of the next layer
                    self._blk_done.record(self.third)
                    if self.g_blk0[nxt] is not None:
                        self.g_blk0[nxt].replay()        # the input layer's block: waits for the end of S, raises FLAG_BLOCK0
                        self._blk0_done.record(self.third)
            with torch.cuda.stream(side):
                self.g_bwd[cur].replay()                 # B: waits for the flag S raises when it starts
                if on_side is not None:
                    on_side()
                # one wait on the critical stream instead of two: "B done" below also means "all blocks of S built" (and the
                # pass complete: the next update may write the rows)
                if self.g_norm is not None or self.g_blk[nxt] is not None:
                    side.wait_event(self._blk_done)
                if self._flag_boundary:
                    _lib.check(_lib.lib.bliss_flag_raise(self.sampler._engine.flags.data_ptr() + 4 * self.FLAG_B_DONE, side.cuda_stream),
                               "bliss_flag_raise")
                self._bwd_done.record(side)
        else:
            self.g_fwd[cur].replay()                     # F + X
            self._fwd_done.record(main)
            with torch.cuda.stream(side):
                side.wait_event(self._fwd_done)
                self.g_bwd[cur].replay()                 # B, beside ...
                if on_side is not None:
                    on_side()
                self._bwd_done.record(side)
            self.g_smp[nxt].replay()                     # ... S (needs the EXP3 weights X just wrote: same stream)

    def _join(self):
        torch.cuda.current_stream().wait_event(self._bwd_done)     # (flag mode: implies the early blocks, see _half)
        if self._defer_block0 and self.graph:
            torch.cuda.current_stream().wait_event(self._blk0_done)

    def _prime_boundary(self):
        # the first forward pass of a run of replays waits for a backward pass nobody launched
        if self.use_flags and self._flag_boundary and self.graph:
            eng = self.sampler._engine
            _lib.check(_lib.lib.bliss_flag_raise(eng.flags.data_ptr() + 4 * self.FLAG_B_DONE, torch.cuda.current_stream().cuda_stream),
                       "bliss_flag_raise")
            if self._defer_block0:                       # (the batch in flight was sampled completely: _join / prime)
                _lib.check(_lib.lib.bliss_flag_raise(eng.flags.data_ptr() + 4 * self.FLAG_BLOCK0, torch.cuda.current_stream().cuda_stream),
                           "bliss_flag_raise")

    def _after_blocks(self):
        """The stream on which 'the sampler of the last half has finished' is to be recorded: the third one when it builds the
        last block (its counts record and error word are complete only then)."""
        return torch.cuda.stream(self.third) if (self._defer_block0 and self.graph) else contextlib.nullcontext()

    def _replay(self, first_chain=False):
        eng = self.sampler._engine
        self._prime_boundary()
        for cur, nxt, chain in ((0, 1, first_chain), (1, 0, True)):
            eng.static_rng_begin(chain)                  # the serial MT19937 chain of S starts now, beside F + X
            self._half(cur, nxt)
            if self._defer_block0:
                torch.cuda.current_stream().wait_event(self._blk0_done)   # (the counts record static_rng_end copies)
            eng.static_rng_end(nxt)
        self._join()

    def __call__(self, loader):
        """Two steps: trains the batch sampled by the previous call and the next batch of ``loader``; samples two."""
        self._load(loader)
        self._replay()
        self._finish_pair()
        return self.losses

    # -- capacities that follow the run --------------------------------------------------------------------------------
    # Static capacities come from a few steps taken while the bandit weights are uniform; as the weights move, the kept sets
    # and blocks drift.  A step that exceeds a capacity is clamped and flagged only afterwards (its update is invalid), so
    # the loop watches the sizes that come back with every pair and acts BEFORE that: once a size passes ``regrow_at`` of
    # its capacity, the next run() first trains the batch in flight, recalibrates from the high-water marks, re-captures
    # the graphs and carries on -- the batches, their order and the sampler's random stream are those of the plain loop.
    regrow_at = 0.85

    def _watch(self, sizes):
        caps = self.sampler._engine.caps
        L = len(caps)
        for sz in sizes:                                   # per batch: blocks input-most first = sampling order reversed
            for l, s_ in enumerate(sz):
                n = L - 1 - l
                hw = self._hw[n]
                for k in ("K", "B", "E"):
                    hw[k] = max(hw.get(k, 0), int(s_[k]))
                if s_["K"] > self.regrow_at * caps[n]["K"] or s_["B"] > self.regrow_at * caps[n]["B"] or \
                        (self.distributed and s_["B"] > self.regrow_at * caps[n].get("X", caps[n]["B"])):
                    self._needs_regrow = True

    def _regrow(self, loader):
        """Train the batch in flight, enlarge the capacities to the high-water marks (x the calibration margins), re-capture.
        Returns the sizes of the two batches the re-capture sampled (it replays one pair)."""
        eng = self.sampler._engine
        L = len(self.sampler.nodes_per_layer)
        self.drain()
        mx = [dict(h) for h in self._hw]
        if self.distributed:                               # every rank must end up with the same capacities
            import torch.distributed as dist
            t = torch.tensor([[m["K"], m["B"], m["E"]] for m in mx], dtype=torch.int64, device=self.g.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mx = [dict(K=int(k), B=int(b), E=int(e)) for k, b, e in t.tolist()]
        GraphedTrainStep.close(self)                       # quiesce, drop the graphs recorded for the old shapes
        fan = [self.sampler.nodes_per_layer[b] for b in reversed(range(L))]
        eng.set_static_caps(self.bs, fan, mx, *self._margins)
        if self.use_flags:
            eng.scratch_sets = max(eng.scratch_sets, L)
        self._needs_regrow = False
        self.regrows = getattr(self, "regrows", 0) + 1
        self.prime(next(loader))
        primed = [dict(S=b._counts.S, E=b._counts.E, C=b._counts.C, K=b._counts.K, B=b._counts.B) for b in reversed(eng._static[0][0])]
        self._capture_graphs(loader)
        return [primed] + self.sizes2()                   # every batch sampled here, in sampling order

    def run(self, loader, n_pairs, ring=8, pair_events=None):
        # The host stays a few pairs ahead of the device; a generation-2 garbage collection in the middle of the loop can take
        # longer than that lead and drain the queues (one pair of > 3 ms in some 400-step windows): not during the loop.
        import gc
        was_enabled = gc.isenabled()
        if was_enabled:
            gc.disable()
        try:
            return self._run(loader, n_pairs, ring, pair_events)
        finally:
            if was_enabled:
                gc.enable()

    def _run(self, loader, n_pairs, ring=8, pair_events=None):
        """``n_pairs`` calls without a host round trip in between: the generator state is chained on the device from
        batch to batch (torch's CPU generator is brought up to date once, at the end), and sizes / error words come back
        through a small ring of pinned buffers while later pairs are already running.  Returns the block sizes of every
        batch sampled, in order.  ``pair_events``: a list that receives one timing event per pair boundary (recorded on
        the critical stream at the start of every pair and after the last one): consecutive differences are the device
        time of two train steps each."""
        eng = self.sampler._engine
        L = len(self.sampler.nodes_per_layer)
        if getattr(self, "_ring", None) is None or len(self._ring) != ring:
            self._ring = [torch.empty(2 * L * 10, dtype=torch.int32).pin_memory() for _ in range(ring)]
            self._ring_ev = [torch.cuda.Event() for _ in range(ring)]
        sizes, pending, bad = [], [], 0
        if getattr(self, "_needs_regrow", False):          # (the re-capture trains three batches itself: one drained, one pair)
            sizes += self._regrow(loader)
            self._watch(sizes)

        def collect(i):
            nonlocal bad
            self._ring_ev[i].synchronize()
            raw = self._ring[i].numpy().tobytes()
            for half in range(2):
                cs = [_lib.LayerCounts.from_buffer_copy(raw[40 * (half * L + n): 40 * (half * L + n) + 40]) for n in range(L)]
                for c in cs:
                    bad |= c.err
                sizes.append([dict(S=c.S, E=c.E, C=c.C, K=c.K, B=c.B) for c in reversed(cs)])

        eng.stage_rng_from_torch()
        # Free-running loop: nothing but graph replays and event records goes onto the main stream.  The generator hand-over
        # between two samplers (state commit, counts to the host, control block of the next generator) is one kernel on the
        # generator's stream (static_rng_chain); the seed ids of later batches are copied on the backward pass's stream.
        main = torch.cuda.current_stream()
        self._prime_boundary()
        if n_pairs:
            self.seeds2[1].copy_(next(loader))           # S(b) runs first, then S(a')
            self.seeds2[0].copy_(next(loader))
        trace = self._host_trace = [] if pair_events is not None else None   # (host clock per pair: enqueue start, time blocked)
        for k in range(n_pairs):
            t_in = time.perf_counter() if trace is not None else 0.0
            while pending and pending[0] <= k - ring:    # this pair reuses that pair's record
                collect(pending.pop(0) % ring)
            if bad:
                # an error word came back with a pair's record (capacity overrun, non-finite weight, ...): every later step
                # builds on invalid blocks / EXP3 rows, so stop enqueueing -- the error surfaces within `ring` pairs of the step
                # that raised it, not at the end of the window (round-2 advice)
                n_done = k
                break
            if pair_events is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(main)
                pair_events.append(ev)
                trace.append((t_in, time.perf_counter() - t_in))
            last = k == n_pairs - 1
            r = self._ring[k % ring]
            for cur, nxt in ((0, 1), (1, 0)):
                if k == 0 and cur == 0:
                    eng.static_rng_begin(False)          # from the host-staged state
                elif cur == 0:                           # ends S(a') of the previous pair; its sizes complete that pair's record
                    with self._after_blocks():
                        eng.static_rng_chain(0, self._ring[(k - 1) % ring][L * 10:])
                else:
                    with self._after_blocks():
                        eng.static_rng_chain(1, r[:L * 10])
                # (no wait for the hand-over on the main stream: the sampler's first random-number wait checks that the control
                # block is the new generator's -- the event round trip cost ~12 us between every two steps)
                if cur == 0 and k > 0:                   # the previous pair's record is complete once that hand-over has run
                    eng.static_rng_record(self._ring_ev[(k - 1) % ring])
                    pending.append(k - 1)
                # the sampler that read slot ``cur``'s seed ids finished before this half's forward pass: load the next batch
                # there, behind the backward pass.  The loader itself works on the main stream (a new epoch shuffles there).
                on_side = None
                if (cur == 0 and k > 0) or (cur == 1 and not last):
                    batch = next(loader)
                    self._seed_ev.record(main)
                    batch.record_stream(self.side)

                    def on_side(c=cur, b=batch):
                        self.side.wait_event(self._seed_ev)
                        self.seeds2[c].copy_(b)
                self._half(cur, nxt, on_side=on_side)
            if last:
                if self._defer_block0:
                    main.wait_event(self._blk0_done)
                eng.static_rng_end(0)
                r[L * 10:].copy_(eng._slot_counts[0], non_blocking=True)
                self._join()
                if pair_events is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(main)
                    pair_events.append(ev)
                self._ring_ev[k % ring].record(main)
                pending.append(k)
        else:
            n_done = n_pairs
        if n_done < n_pairs:                             # stopped early: let the device finish what is enqueued, then report
            torch.cuda.synchronize()                     # (the loop object is not usable afterwards: results are invalid anyway)
            for i in pending:
                collect(i % ring)
            raise RuntimeError(f"static-shape step exceeded its capacities or hit a kernel error 0x{bad:x} ({_lib.err_string(bad)}) "
                               f"within the last {ring} pairs before pair {n_done} of {n_pairs}; results from there on are invalid")
        for i in pending:
            collect(i % ring)
        if n_pairs:
            # what finish_static(1) reads: the last pair's first sampler handed its sizes to the ring
            eng._slot_counts_host[1].copy_(self._ring[(n_pairs - 1) % ring][:L * 10])
            self._finish_pair()
        if bad:
            raise RuntimeError(f"static-shape step exceeded its capacities or hit a kernel error 0x{bad:x} "
                               f"({_lib.err_string(bad)}) before the early-warning regrow could act; results are invalid -- "
                               f"raise the margins or lower regrow_at")
        self._watch(sizes)
        if self.distributed and hasattr(self.sampler, "check_errors"):
            # the exchange truncates an update list that outgrew its capacity and only flags it on the sampler: surface
            # it with the call that produced it, not at the end of training (one tiny read-back per run(), not per step)
            self.sampler.check_errors()
        return sizes

    def eager_pair(self, loader):
        """The same two steps launched kernel by kernel (used to time individual kernels)."""
        self._load(loader)
        self.losses = self._pair()
        self._finish_pair()
        return self.losses

    def drain(self):
        """Train on the batch that is sampled but not trained yet (end of training)."""
        main, side = torch.cuda.current_stream(), self.side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            loss = self._backward(self._forward(self.mfgs[0]))
        main.wait_stream(side)
        if self._defer:
            self.sampler._settle()
        main.synchronize()
        self.num_steps += 1
        return loss

    def _graph_attrs(self):
        return ("graph", "g_main", "g_fwd", "g_bwd", "g_smp", "g_blk", "g_blk0", "g_norm")

    def close(self):
        """Train the batch still in flight (``drain``), wait for every stream of the loop, then destroy the graphs."""
        if self.graph and self.mfgs[0] is not None:
            self.drain()
        for st in (self.side, self.third):
            st.synchronize()
        super().close()
        self.mfgs = [None, None]
        if self._defer:
            self.sampler.enable_deferred_normalize(False)
            self._defer = False

    def sizes2(self):
        """sizes() for each of the two batches sampled by the last call."""
        return [[dict(S=c.S, E=c.E, C=c.C, K=c.K, B=c.B) for c in reversed(cs)] for cs in self.last_counts2]
