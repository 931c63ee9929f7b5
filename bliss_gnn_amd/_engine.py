"""Host-side driver of the gfx950 sampler kernels.

One ``sample_blocks`` call = L x (frontier_prob, mt19937, poisson_select, build_block) enqueued on
the current stream with NO host round trip in between: every size (S, E, C, K, B) stays on the
device and the next layer reads its seed count from the previous layer's counts record.  A single
device->host copy at the end returns all sizes, the error words and the advanced generator state
(the reference syncs >= 24 times per step).

Buffers are sized by per-layer capacities; if a capacity is exceeded the kernels clamp, flag it,
and the whole call is repeated with larger buffers from the same generator snapshot (steady state:
never).  PyTorch is used for device memory and the current stream only.
"""
import contextlib
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .graph import Block, Graph

_CHUNK = 1024
_CAP_ERRS = 1 | 2 | 4 | 8 | 64      # (128, the random stream, is fatal: the stream is sized for the worst case)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _up8(n):
    return (int(n) + 7) & ~7


class _LayerWs:
    """Persistent per-layer scratch (internal to the sampler, reused every call)."""

    def __init__(self, dev, cap_s, cap_c):
        self.cap_s, self.cap_c = cap_s, cap_c
        self.seg_ptr = torch.empty(cap_s + 1, dtype=torch.int32, device=dev)
        self.seed_acc = torch.empty(60 * cap_s + 64, dtype=torch.uint8, device=dev)    # 7 x u64 per seed + k_seg_scan's long-column list
        self.cand_nid = torch.empty(cap_c, dtype=torch.int32, device=dev)
        self.new_id = torch.empty(cap_c, dtype=torch.int32, device=dev)
        self.p = torch.empty(cap_c, dtype=torch.bfloat16, device=dev)
        self.P = torch.empty(cap_c, dtype=torch.bfloat16, device=dev)
        self.uniforms = torch.empty(cap_c, dtype=torch.float32, device=dev)     # explicit-uniforms path only
        self.src_cnt = None


class LayerEngine:
    """Per-graph sampling state.  Not re-entrant and not shareable between DataLoader workers --
    the same ownership rule as the reference sampler's mutable state (bandit_sampler.py:43)."""

    def __init__(self, g: Graph):
        if g.device.type != "cuda":
            raise RuntimeError("bliss_gnn_amd samples on the GPU only; move the graph with g.to('cuda') "
                               "(the reference's --data-cpu/--use-uva modes are out of scope)")
        self.g = g
        dev = g.device
        V = g.num_nodes()
        self.V, self.Eg = V, g.num_edges()
        if self.Eg >= 2 ** 31:
            raise RuntimeError("int32 edge positions: graphs with >= 2^31 edges are not supported")
        # Scratch shared by the layers exists ``scratch_sets`` times, layer n (sampling order) uses set n % scratch_sets: the
        # block passes of layer n touch none of what the candidate pipeline of layer n + 1 touches, so a caller may run them
        # beside each other on two streams (train.PipelinedTrainStep does, with one set per layer).
        self.scratch_sets = 2
        self._sets = {}
        self.first_pos = torch.full((V,), -1, dtype=torch.int32, device=dev)      # 0xFFFFFFFF
        self.acc_p2 = torch.zeros(V, dtype=torch.int64, device=dev)
        self.c_graph = _lib.Graph(g.indptr.data_ptr(), g.indices.data_ptr(), _ptr(g.eid), V, self.Eg)
        self.flags = torch.zeros(16, dtype=torch.int32, device=dev)              # cross-stream hand-offs (bliss_flag_wait)
        self.fs_ticket = torch.zeros(1, dtype=torch.int32, device=dev)           # bliss_layer_ws_t::fs_ticket
        self.fuse_scale = os.environ.get("BLISS_FUSE_SCALE", "1") != "0"
        self.flag_err = torch.zeros(1, dtype=torch.int32, device=dev)
        # binned candidate pipeline (csrc/sampler.hip): LDS-resident per-source reductions when |V| / n_bins slots fit in
        # 64 KiB; otherwise (or with BLISS_BINS=0) the memory-side atomic passes.  Scratch shared by all layers.
        self.n_bins = 0
        for nb in ((int(os.environ["BLISS_NBINS"]),) if os.environ.get("BLISS_NBINS") else (256, 1024)):
            if -(-V // nb) * 12 <= 64 * 1024:
                self.n_bins = nb
                break
        if os.environ.get("BLISS_BINS", "1") == "0" or self.Eg > 2048 * 4096 * 32:     # (bitmap tiles: MAX_TILES in sampler.hip)
            self.n_bins = 0
        self._bins = None
        self.hist = torch.zeros(32768, dtype=torch.int32, device=dev)      # self-cleaning (k_poisson_scale)
        self.mt_dev = torch.empty(626, dtype=torch.int32, device=dev)
        self.mt_host = torch.empty(626, dtype=torch.int32).pin_memory()        # state handed to the device
        self.mt_back = torch.empty(626, dtype=torch.int32).pin_memory()        # state handed back
        self._static = {}
        self._slot_bufs, self._slot_counts, self._slot_counts_host = {}, {}, {}
        self.caps = None
        self.ws = None
        self.counts_host = None
        self.retries = 0

    def _set(self, n):
        """The shared-scratch set of layer n: dense node maps (-1 = clean), span table, chunk counters, spill buffers."""
        i = n % self.scratch_sets
        if i not in self._sets:
            dev, V = self.g.device, self.V
            local_id = torch.full((V,), -1, dtype=torch.int32, device=dev)
            self._sets[i] = dict(local_id=local_id, kept_map=torch.full((V,), -1, dtype=torch.int32, device=dev),
                                 span_seg=torch.zeros(self.Eg // 256 + 2, dtype=torch.int32, device=dev),
                                 chunk_cnt=torch.empty(max(self.Eg, V) // _CHUNK + 2, dtype=torch.int32, device=dev),
                                 kept_rec=None, span_cnt=None,
                                 c_maps=_lib.NodeMaps(local_id.data_ptr(), self.first_pos.data_ptr(), self.acc_p2.data_ptr()))
        return self._sets[i]

    # ------------------------------------------------------------------ capacities
    def _init_caps(self, S0, fanouts_sampling_order):
        caps, s = [], int(S0)
        for f in fanouts_sampling_order:
            k = min(self.V, int(1.25 * (f + s)) + 64)
            caps.append(dict(S=s, C=self.V, K=k, B=int(min(self.Eg, max(1 << 16, 48 * k)))))
            s = k
        return caps

    def _ensure(self, S0, fan):
        if self.caps is None or len(self.caps) != len(fan):
            self.caps = self._init_caps(S0, fan)
            self.ws = None
        elif self.caps[0]["S"] < S0:
            fresh = self._init_caps(S0, fan)
            self.caps = [{k: max(a[k], b[k]) for k in a} for a, b in zip(self.caps, fresh)]
            self.ws = None
        if self.ws is None:
            dev = self.g.device
            self.ws = [_LayerWs(dev, c["S"], c["C"]) for c in self.caps]
            self.counts_host = torch.empty(len(fan) * 10, dtype=torch.int32).pin_memory()
            # one random stream per call: the layers consume consecutive slices of it
            self.rng_cap = sum(c["C"] for c in self.caps)
            # (jump-ahead tables for this capacity: the stream is then generated by several workgroups at once, csrc/rng.hip)
            plan = (C.c_int32 * 4)()
            _lib.check(_lib.lib.bliss_rng_prepare(self.rng_cap, plan), "bliss_rng_prepare")
            self.rng_plan = tuple(plan)
            n_words = 624 * (max(plan[3], self.rng_cap // 624 + 1) + 2)
            self.rng_out = torch.empty(n_words, dtype=torch.float32, device=dev)
            self.rng_raw = torch.empty(n_words, dtype=torch.int32, device=dev)
            self.rng_ctl = torch.zeros(8 + len(fan), dtype=torch.int32, device=dev)     # ctl[8] + per-layer offsets

    def _bin_buffers(self):
        if self._bins is None:
            dev, nb = self.g.device, self.n_bins
            cap = int(1.25 * self.Eg / nb) + 8192
            words = (-(-self.Eg // 4096) + 1) * 128 + 4
            self._bins = dict(cap=cap, cursor=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                              rec=torch.empty(nb * cap, dtype=torch.int64, device=dev),
                              bitmap=torch.zeros(words, dtype=torch.int32, device=dev),
                              prefix=torch.empty(2048 + words, dtype=torch.int32, device=dev),
                              tkey=torch.empty(self.V, dtype=torch.int64, device=dev),
                              tsum=torch.empty(self.V, dtype=torch.int64, device=dev))
        return self._bins

    def _grow(self, errs):
        for n, e in enumerate(errs):
            c = self.caps[n]
            if e & 2 and self.n_bins:       # a bin overflowed (extremely skewed source ids): use the atomic passes
                self.n_bins = 0
            if e & 64:
                c["S"] = min(self.V, 2 * c["S"])
            if e & 4:
                c["K"] = min(self.V, 2 * c["K"])
            if e & 8:
                c["B"] = int(min(self.Eg, 2 * c["B"]))
        for n in range(len(self.caps) - 1):                       # a layer's seeds are the previous layer's kept nodes
            self.caps[n + 1]["S"] = max(self.caps[n + 1]["S"], self.caps[n]["K"])
        self.ws = None
        self.retries += 1

    # ------------------------------------------------------------------ torch CPU generator <-> device
    @staticmethod
    def _rng_fields(state_bytes):
        a = state_bytes.numpy()
        return a[8:12].view(np.int32), a[16:24].view(np.int64), a[24:24 + 624 * 8].view(np.uint64)

    def _stage_rng(self, snapshot):
        """Host half of the hand-over: put the generator state where the (possibly graph-captured) H2D copy reads it."""
        left, nxt, st = self._rng_fields(snapshot)
        h = self.mt_host.numpy()
        h[:624] = st.astype(np.uint32).view(np.int32)
        h[624], h[625] = int(left[0]), int(nxt[0])

    def _commit_rng(self, snapshot):
        """Make the global CPU generator continue after the numbers the device consumed."""
        left, nxt, st = self._rng_fields(snapshot)
        h = self.mt_back.numpy()
        st[:] = h[:624].view(np.uint32).astype(np.uint64)
        left[0], nxt[0] = int(h[624]), int(h[625])
        torch.set_rng_state(snapshot)

    # ------------------------------------------------------------------ one sample_blocks call
    def sample_blocks(self, w_rows, seeds, fanouts, mode, eta, eps=0.9999, uniforms=None):
        """The ``for block_id in reversed(range(L))`` loop of bandit_sampler.py:350-366 /
        ladies_sampler.py:112-122.  ``w_rows[n]`` / ``fanouts[n]`` are in SAMPLING order (last block
        first).  Returns the blocks in sampling order.

        ``uniforms``: optional list of fp32 vectors replacing the draws from torch's global CPU
        generator (the stream ATen's CPU ``torch.bernoulli`` consumes)."""
        seeds = seeds.to(torch.int32).contiguous()
        L = len(fanouts)
        self._ensure(int(seeds.numel()), fanouts)
        snapshot = torch.get_rng_state() if uniforms is None else None
        while True:
            out = self._enqueue(w_rows, seeds, fanouts, mode, eta, eps, uniforms, snapshot)
            counts_dev = out[0]
            self.counts_host.copy_(counts_dev, non_blocking=True)
            if snapshot is not None:
                self.mt_back.copy_(self.mt_dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()                 # the ONE sync of the step's sampling
            raw = self.counts_host.numpy().tobytes()
            cnts = [_lib.LayerCounts.from_buffer_copy(raw[40 * n: 40 * n + 40]) for n in range(L)]
            errs = [c.err for c in cnts]
            bad = 0
            for e in errs:
                bad |= e
            if bad & ~_CAP_ERRS:
                raise RuntimeError(f"sampler kernel error 0x{bad:x}: {_lib.err_string(bad)}")
            if bad & 1:
                raise RuntimeError("frontier longer than the graph has edges (repeated seeds?) or >= 2^31 positions")
            if bad & 2 and not self.n_bins:
                raise RuntimeError("candidate capacity exceeded / seed id out of range")
            if bad == 0:
                break
            self._grow(errs)
            self._ensure(int(seeds.numel()), fanouts)
        if snapshot is not None:
            self._commit_rng(snapshot)
        return self._finish(out, cnts)

    # ------------------------------------------------------------------ non-Poisson samplers (multinomial draw)
    def sample_blocks_multinomial(self, w_rows, seeds, fanouts, mode, eta, replace=False, fp32_importance=False):
        """BanditLadiesSampler / LadiesSampler (bandit_sampler.py:84-99, ladies_sampler.py:54-69): the node importances
        are computed on the device; the draw is ``torch.multinomial`` itself, on the host, on those bits (ATen's CPU
        kernel takes its exponentials from an MKL stream seeded by the global generator -- there is nothing to restate),
        so this path syncs once per layer like the reference does.  ``fp32_importance``: hand the draw fp32 importances
        (LadiesSampler's non-importance branch builds fp32 ones, ladies_sampler.py:50; ATen draws in the input's dtype)."""
        seeds = seeds.to(torch.int32).contiguous()
        L = len(fanouts)
        self._ensure(int(seeds.numel()), fanouts)
        # a multinomial draw keeps min(num, C) nodes plus the seeds: make room
        for n in range(L):
            need = min(self.V, fanouts[n] + self.caps[n]["S"] + 64)
            if self.caps[n]["K"] < need:
                self.caps[n]["K"] = need
            if n + 1 < L and self.caps[n + 1]["S"] < self.caps[n]["K"]:
                self.caps[n + 1]["S"] = self.caps[n]["K"]
                self.ws = None
        self._ensure(int(seeds.numel()), fanouts)
        while True:
            dev, st = self.g.device, _stream()
            counts = torch.empty(L * 10, dtype=torch.int32, device=dev)
            eta_f, ome_f = float(np.float32(eta)), float(np.float32(1.0 - eta))
            layers, state = [], (seeds, int(seeds.numel()), 0)
            for n in range(L):
                cur_seeds, n_seeds, n_seeds_dev = state
                c_ws, c_out, lay, cnt_ptr, kept_nid = self._layer_buffers(n, counts)
                w_pos = w_rows[n]
                _lib.check(_lib.lib.bliss_frontier_prob(C.byref(self.c_graph), C.byref(self._set(n)["c_maps"]), w_pos.data_ptr(),
                                                        cur_seeds.data_ptr(), n_seeds, n_seeds_dev, self.caps[n]["S"], mode,
                                                        eta_f, ome_f, self.Eg, C.byref(c_ws), st), "bliss_frontier_prob")
                self.counts_host.copy_(counts, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                Cn = int(self.counts_host[10 * n + 2])
                prob = self.ws[n].p[:Cn].cpu()
                if fp32_importance:
                    prob = prob.float()
                chosen = torch.multinomial(prob, min(int(fanouts[n]), Cn), replacement=replace)        # bandit_sampler.py:98
                chosen_dev = chosen.to(torch.int32).to(dev)
                _lib.check(_lib.lib.bliss_multinomial_select(C.byref(c_ws), chosen_dev.data_ptr(), int(chosen_dev.numel()), st),
                           "bliss_multinomial_select")
                _lib.check(_lib.lib.bliss_build_block(C.byref(self.c_graph), C.byref(self._set(n)["c_maps"]), w_pos.data_ptr(),
                                                      cur_seeds.data_ptr(), self.caps[n]["S"], mode, eta_f, ome_f, self.Eg,
                                                      C.byref(c_ws), C.byref(c_out), st), "bliss_build_block")
                layers.append(lay)
                state = (kept_nid, -1, cnt_ptr + 12)
            self.counts_host.copy_(counts, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            raw = self.counts_host.numpy().tobytes()
            cnts = [_lib.LayerCounts.from_buffer_copy(raw[40 * n: 40 * n + 40]) for n in range(L)]
            bad = 0
            for c in cnts:
                bad |= c.err
            if bad & ~_CAP_ERRS or bad & 2:
                raise RuntimeError(f"sampler kernel error 0x{bad:x}: {_lib.err_string(bad)}")
            if bad == 0:
                return self._finish((counts, layers), cnts)
            self._grow([c.err for c in cnts])
            self._ensure(int(seeds.numel()), fanouts)

    def _layer_buffers(self, n, counts, slot=None):
        """Caller-owned outputs of layer n (one int32 and one bf16 allocation, sliced) + the C descriptors.
        ``slot``: static-shape callers keep one persistent set of outputs per slot (a pipelined train loop holds two
        batches at a time); None = fresh memory per call."""
        dev = self.g.device
        cap, ws = self.caps[n], self.ws[n]
        cs, ck, cb = cap["S"], cap["K"], cap["B"]
        build_t = cs <= 32768                     # by-source index built by the sampler kernels themselves
        if ws.src_cnt is None or ws.src_cnt.numel() < ck + 1:
            ws.src_cnt = torch.empty(ck + 1, dtype=torch.int32, device=dev)
        key = (slot, n, cs, ck, cb)
        if slot is not None and key in self._slot_bufs:
            ibuf, hbuf = self._slot_bufs[key]
        else:
            ibuf = torch.empty(_up8(cs + 1) + 4 * _up8(cb) + _up8(ck) + (2 * _up8(cb) + _up8(ck + 1) if build_t else 0),
                               dtype=torch.int32, device=dev)
            hbuf = torch.empty(2 * _up8(cb) + _up8(ck), dtype=torch.bfloat16, device=dev)
            if slot is not None:
                self._slot_bufs[key] = (ibuf, hbuf)
        o = 0
        b_indptr = ibuf[o:o + cs + 1]; o += _up8(cs + 1)
        b_src = ibuf[o:o + cb]; o += _up8(cb)
        b_dst = ibuf[o:o + cb]; o += _up8(cb)
        b_pos = ibuf[o:o + cb]; o += _up8(cb)
        b_eid = ibuf[o:o + cb]; o += _up8(cb)
        kept_nid = ibuf[o:o + ck]; o += _up8(ck)
        t_indptr = t_edge = t_scr = None
        if build_t:
            t_indptr = ibuf[o:o + ck + 1]; o += _up8(ck + 1)
            t_edge = ibuf[o:o + cb]; o += _up8(cb)
            t_scr = ibuf[o:o + cb]
        b_w = hbuf[0:cb]
        b_q = hbuf[_up8(cb):_up8(cb) + cb]
        node_prob = hbuf[2 * _up8(cb):2 * _up8(cb) + ck]
        cnt_ptr = counts.data_ptr() + 40 * n
        sset = self._set(n)                       # which set of shared scratch this layer uses
        c_ws = _lib.LayerWs(cnt_ptr, ws.seg_ptr.data_ptr(), ws.seed_acc.data_ptr(), sset["chunk_cnt"].data_ptr(),
                            ws.cand_nid.data_ptr(), ws.p.data_ptr(), ws.P.data_ptr(), ws.new_id.data_ptr(),
                            kept_nid.data_ptr(), node_prob.data_ptr(), self.hist.data_ptr(),
                            ws.src_cnt.data_ptr() if build_t else 0, cap["C"], ck)
        c_ws.kept_map, c_ws.span_seg = sset["kept_map"].data_ptr(), sset["span_seg"].data_ptr()
        e_bound = max(c.get("E", self.Eg) for c in self.caps)        # one buffer for all layers (descriptors of earlier layers stay valid)
        if e_bound <= (1 << 24):        # spill buffer for the block passes: 16 B per frontier position, only for bounded frontiers
            pos = -(-e_bound // 1024) * 1024
            if sset["kept_rec"] is None or sset["kept_rec"].numel() < pos * 2:
                sset["kept_rec"] = torch.empty(pos * 2, dtype=torch.int64, device=dev)
                sset["span_cnt"] = torch.zeros(pos // 256 + 4, dtype=torch.int32, device=dev)
            kr, sc = sset["kept_rec"], sset["span_cnt"]
            c_ws.kept_rec, c_ws.span_cnt, c_ws.kept_rec_positions = kr.data_ptr(), sc.data_ptr(), kr.numel() // 2
        if self.n_bins:
            b = self._bin_buffers()
            c_ws.n_bins, c_ws.bin_cap = self.n_bins, b["cap"]
            c_ws.bin_cursor, c_ws.bin_rec = b["cursor"].data_ptr(), b["rec"].data_ptr()
            c_ws.bitmap, c_ws.word_prefix = b["bitmap"].data_ptr(), b["prefix"].data_ptr()
            c_ws.touched_key, c_ws.touched_sum = b["tkey"].data_ptr(), b["tsum"].data_ptr()
        c_out = _lib.BlockOut(b_indptr.data_ptr(), b_src.data_ptr(), b_dst.data_ptr(), b_pos.data_ptr(), b_eid.data_ptr(),
                              b_w.data_ptr(), b_q.data_ptr(), _ptr(t_indptr), _ptr(t_edge), _ptr(t_scr), cb)
        lay = (b_indptr, b_src, b_dst, b_pos, b_eid, b_w, b_q, kept_nid, node_prob, counts[10 * n:10 * n + 10], t_indptr, t_edge)
        return c_ws, c_out, lay, cnt_ptr, kept_nid

    # ------------------------------------------------------------------ static-shape (graph-capturable) variant
    def set_static_caps(self, S0, fanouts, max_sizes, k_margin=1.5, b_margin=3.0):
        """Fix the capacities from sizes observed in exact mode (``max_sizes[n] = dict(K=, B=)`` in sampling
        order).  Static shapes mean no host round trip inside a step and a step that can be captured in a HIP graph;
        the price is that a step whose true sizes exceed these capacities is detected only afterwards."""
        caps, s = [], int(S0)
        for n, f in enumerate(fanouts):
            k = min(self.V, int(k_margin * max_sizes[n]["K"]) + 256)
            b = int(min(self.Eg, int(b_margin * max_sizes[n]["B"]) + 4096))
            # E only sizes launch grids (every kernel strides over the true count): a tight bound lets the hardware balance
            # the workgroups instead of a capped grid looping unevenly
            e = int(min(self.Eg, int(1.5 * max_sizes[n].get("E", self.Eg)) + 65536))
            # X: how many block edges a multi-GPU step EXCHANGES per block (update lists are sent capacity-sized; B's margin
            # is for memory that costs nothing, X's for bytes that cross xGMI every step); a longer list is flagged
            x = int(min(b, int(1.5 * max_sizes[n]["B"]) + 2048))
            caps.append(dict(S=s, C=self.V, K=k, B=b, E=e, X=x))
            s = k
        self.caps, self.ws = caps, None
        self._ensure(S0, fanouts)

    def stage_rng_from_torch(self):
        """Host side, before enqueueing / replaying a static step: snapshot torch's CPU generator for the device."""
        self._static_snapshot = torch.get_rng_state()
        self._stage_rng(self._static_snapshot)

    def static_rng_begin(self, chain_rng=False):
        """Start the random-number generator of the NEXT enqueue_static(..., external_rng=True) now, on the current stream:
        a pipelined loop calls this before the sampler's other inputs are ready, so the serial MT19937 chain is off the
        critical path.  Plain launches (not graph-captured): the generator runs on the library's own stream."""
        if not chain_rng:
            self.mt_dev.copy_(self.mt_host, non_blocking=True)
        _lib.check(_lib.lib.bliss_rng_stream_begin(self.mt_dev.data_ptr(), self.rng_ctl.data_ptr(), self.rng_out.data_ptr(),
                                                   self.rng_raw.data_ptr(), self.rng_cap, _stream()), "bliss_rng_stream_begin")

    def static_rng_end(self, slot=0):
        """Join the generator started by static_rng_begin and leave the state after exactly sum(C) draws in mt_dev / mt_back."""
        _lib.check(_lib.lib.bliss_rng_stream_end(self.mt_dev.data_ptr(), self.rng_ctl.data_ptr(), self.rng_raw.data_ptr(),
                                                 self.rng_cap, self._slot_counts[slot].data_ptr() + 20, _stream()), "bliss_rng_stream_end")
        self.mt_back.copy_(self.mt_dev, non_blocking=True)
        self._slot_counts_host[slot].copy_(self._slot_counts[slot], non_blocking=True)

    def static_rng_chain(self, slot, counts_host):
        """static_rng_end(slot) + static_rng_begin(chain_rng=True) without touching the current stream: the hand-over runs on
        the generator's own stream (bliss_rng_stream_chain), ordered after what the current stream holds now (the sampler of
        ``slot``); the counts records of ``slot`` arrive in ``counts_host`` (a pinned int32 tensor or a view of one).  Call
        static_rng_ready() before enqueueing the next sampler.  The generator state is not handed back to the host: finish a
        chain with static_rng_end."""
        cd = self._slot_counts[slot]
        assert counts_host.is_pinned() and counts_host.numel() >= cd.numel() and counts_host.dtype == torch.int32
        _lib.check(_lib.lib.bliss_rng_stream_chain(self.mt_dev.data_ptr(), self.rng_ctl.data_ptr(), self.rng_out.data_ptr(),
                                                   self.rng_raw.data_ptr(), self.rng_cap, cd.data_ptr(), cd.numel(),
                                                   counts_host.data_ptr(), _stream()), "bliss_rng_stream_chain")

    def static_rng_record(self, event):
        """Record ``event`` on the generator's stream, i.e. behind the hand-over kernel of the last static_rng_chain (which
        wrote that call's counts_host) and the generator it started."""
        if getattr(self, "_gen_stream", None) is None:
            h = int(_lib.lib.bliss_rng_stream_handle())
            if not h:
                raise RuntimeError("bliss_rng_stream_handle failed")
            self._gen_stream = torch.cuda.ExternalStream(h)
        event.record(self._gen_stream)

    def static_rng_ready(self):
        _lib.check(_lib.lib.bliss_rng_stream_ready(_stream()), "bliss_rng_stream_ready")

    def enqueue_static(self, w_rows, seeds, fanouts, mode, eta, eps=0.9999, slot=0, chain_rng=False, external_rng=False, part=None,
                       w_pend=None, last_block=True, ready_flag=0):
        """Enqueue one sample_blocks on the current stream with capacity-padded outputs and NO sync.  Returns the
        blocks (sampling order); sizes, errors and the generator state are read back by finish().

        ``slot``: which persistent set of output buffers to fill.  ``chain_rng``: continue from the generator state the
        previous enqueue left on the device instead of the host-staged one (several batches sampled per host round trip).
        ``part``: None = the whole call.  "main" = everything except the blocks of all but the last-sampled layer, and layer
        n raises ``flags[n]`` when it starts; "early_blocks" = only those blocks, each behind a bliss_flag_wait on
        ``flags[n + 1]`` -- to be enqueued on ANOTHER stream, with ``scratch_sets`` >= the number of layers (block n then shares
        no scratch with any later layer) and external_rng.  The caller orders the next "main" after both parts.
        ``last_block=False`` with "main": the last-sampled layer's block is left out as well and ``flags[L]`` is raised at the end
        (its draw is done); part "last_block" = only that block, behind a bliss_flag_wait on ``flags[L]``; ``ready_flag``
        (address of a device flag) is raised as soon as the block's forward arrays are final, before its by-source index.
        ``w_pend``: per layer (sampling order) the address of the row's pending-norm word (bliss_exp3_step_deferred)."""
        if part is not None and (self.scratch_sets < len(fanouts) or not external_rng):
            raise ValueError("split enqueue needs one scratch set per layer and an external generator")
        L = len(fanouts)
        out = self._enqueue(w_rows, seeds, fanouts, mode, eta, eps, None, True, slot=slot, chain_rng=chain_rng,
                            external_rng=external_rng, part=part, w_pend=w_pend, last_block=last_block, ready_flag=ready_flag)
        counts_dev, layers = out
        if slot not in self._slot_counts_host:
            self._slot_counts_host[slot] = torch.empty(L * 10, dtype=torch.int32).pin_memory()
        if not external_rng:                     # (external: static_rng_end copies both, off the consumer's critical path)
            self._slot_counts_host[slot].copy_(counts_dev, non_blocking=True)
            self.mt_back.copy_(self.mt_dev, non_blocking=True)
        blocks = []
        for n, lay in enumerate(layers):
            b_indptr, b_src, b_dst, b_pos, b_eid, b_w, b_q, kept_nid, node_prob, cdev, t_indptr, t_edge = lay
            cap = self.caps[n]
            blk = Block(self.g, cap["K"], cap["S"], b_indptr, b_src, b_dst, b_pos, b_eid, kept_nid)
            blk._edge_weights, blk._q, blk._node_prob = b_w, b_q, node_prob
            blk._counts, blk._counts_dev = None, cdev
            blk._nnz_ptr = counts_dev.data_ptr() + 40 * n + 16
            blk._xcap = int(cap.get("X", cap["B"]))
            if t_indptr is not None:
                blk._transposed = (t_indptr, t_edge)
            blk._trace = {}
            blocks.append(blk)
        self._static[slot] = (blocks, L)    # kept: a captured graph replays this enqueue without re-running it
        return blocks

    def finish(self, slot=0, commit=True):
        """After the stream has been synchronised: true sizes, error check, generator hand-back."""
        blocks, L = self._static[slot]
        raw = self._slot_counts_host[slot].numpy().tobytes()
        cnts = [_lib.LayerCounts.from_buffer_copy(raw[40 * n: 40 * n + 40]) for n in range(L)]
        bad = 0
        for c in cnts:
            bad |= c.err
        for b, c in zip(blocks, cnts):
            b._counts = c
        if commit:
            self._commit_rng(self._static_snapshot)
        if bad:
            raise RuntimeError(f"static-shape step exceeded its capacities or hit a kernel error 0x{bad:x} "
                               f"({_lib.err_string(bad)}); the step's results are invalid -- raise the margins")
        return cnts

    def _enqueue(self, w_rows, seeds, fanouts, mode, eta, eps, uniforms, snapshot, slot=None, chain_rng=False, external_rng=False,
                 part=None, w_pend=None, last_block=True, ready_flag=0):
        dev, st = self.g.device, _stream()
        L = len(fanouts)
        if snapshot is not None and not chain_rng and not external_rng:
            if snapshot is not True:
                self._stage_rng(snapshot)
            self.mt_dev.copy_(self.mt_host, non_blocking=True)
        if slot is None:
            counts = torch.empty(L * 10, dtype=torch.int32, device=dev)
        else:
            if slot not in self._slot_counts or self._slot_counts[slot].numel() != L * 10:
                self._slot_counts[slot] = torch.empty(L * 10, dtype=torch.int32, device=dev)
            counts = self._slot_counts[slot]
        use_rng = uniforms is None
        if use_rng and not external_rng:      # fork the generator: it runs beside everything below
            _lib.check(_lib.lib.bliss_rng_stream_begin(self.mt_dev.data_ptr(), self.rng_ctl.data_ptr(), self.rng_out.data_ptr(),
                                                       self.rng_raw.data_ptr(), self.rng_cap, st), "bliss_rng_stream_begin")
        eta_f = float(np.float32(eta))
        ome_f = float(np.float32(1.0 - eta))
        layers = []
        cur_seeds, n_seeds, n_seeds_dev = seeds, int(seeds.numel()), 0
        for n in range(L):
            cap = self.caps[n]
            cs, ws = cap["S"], self.ws[n]
            c_ws, c_out, lay, cnt_ptr, kept_nid = self._layer_buffers(n, counts, slot)
            w_pos = w_rows[n]
            last = n == L - 1
            if w_pend is not None:
                c_ws.w_pend = int(w_pend[n])
            if part == "main":                  # layer n raises flag n when it starts (= everything before it has completed)
                c_ws.entry_flag = self.flags.data_ptr() + 4 * n
            if part in (None, "main"):          # candidates, probabilities, draw: all the next layer needs (its seeds = kept_nid)
                if self.n_bins and self.fuse_scale:
                    # the Poisson scale rides in the last workgroup of the candidate numbering (one launch less per layer)
                    c_ws.fs_ticket = self.fs_ticket.data_ptr()
                    c_ws.fs_fanout, c_ws.fs_eps = int(fanouts[n]), float(eps)
                    if use_rng:
                        c_ws.fs_rng_ctl, c_ws.fs_layer_off = self.rng_ctl.data_ptr(), self.rng_ctl.data_ptr() + 4 * (8 + n)
                        c_ws.fs_is_last, c_ws.fs_rng_cap = int(n == L - 1), self.rng_cap
                _lib.check(_lib.lib.bliss_frontier_prob(C.byref(self.c_graph), C.byref(self._set(n)["c_maps"]), w_pos.data_ptr(),
                                                        cur_seeds.data_ptr(), n_seeds, n_seeds_dev, cs, mode, eta_f, ome_f,
                                                        cap.get("E", self.Eg), C.byref(c_ws), st), "bliss_frontier_prob")
                if use_rng:
                    off_ptr = self.rng_ctl.data_ptr() + 4 * (8 + n)
                    _lib.check(_lib.lib.bliss_poisson_select(C.byref(c_ws), int(fanouts[n]), float(eps), self.rng_out.data_ptr(),
                                                             off_ptr, self.rng_ctl.data_ptr(), int(n == L - 1), self.rng_cap,
                                                             cap["C"], st), "bliss_poisson_select")
                else:
                    u = uniforms[n].to(dev, torch.float32).reshape(-1)
                    m = min(u.numel(), cap["C"])
                    ws.uniforms[:m].copy_(u[:m])
                    _lib.check(_lib.lib.bliss_poisson_select(C.byref(c_ws), int(fanouts[n]), float(eps), ws.uniforms.data_ptr(),
                                                             0, 0, 0, 0, cap["C"], st), "bliss_poisson_select")
            # the block of this layer: nothing the next layer's candidate pipeline reads or writes (own scratch set)
            if part is None or (part == "main" and last and last_block) or (part == "early_blocks" and not last) or \
                    (part == "last_block" and last):
                if part == "last_block" and ready_flag:
                    c_ws.block_ready_flag = int(ready_flag)
                if part in ("early_blocks", "last_block"):   # on another stream: wait until layer n + 1 has started (the last
                    # layer: until the "main" part has raised flags[L]), i.e. layer n's draw is done
                    _lib.check(_lib.lib.bliss_flag_wait(self.flags.data_ptr() + 4 * (n + 1), self.flag_err.data_ptr(), st), "bliss_flag_wait")
                _lib.check(_lib.lib.bliss_build_block(C.byref(self.c_graph), C.byref(self._set(n)["c_maps"]), w_pos.data_ptr(),
                                                      cur_seeds.data_ptr(), cs, mode, eta_f, ome_f, cap.get("E", self.Eg), C.byref(c_ws),
                                                      C.byref(c_out), st), "bliss_build_block")
            layers.append(lay)
            cur_seeds, n_seeds, n_seeds_dev = kept_nid, -1, cnt_ptr + 12          # next layer: S = this layer's K
        if part == "main" and not last_block:
            _lib.check(_lib.lib.bliss_flag_raise(self.flags.data_ptr() + 4 * L, st), "bliss_flag_raise")
        if use_rng and not external_rng:      # join; mt_dev = generator state after exactly sum(C) draws
            _lib.check(_lib.lib.bliss_rng_stream_end(self.mt_dev.data_ptr(), self.rng_ctl.data_ptr(), self.rng_raw.data_ptr(),
                                                     self.rng_cap, counts.data_ptr() + 20, _stream()), "bliss_rng_stream_end")
        return counts, layers

    def _finish(self, out, cnts):
        _, layers = out
        blocks = []
        for n, (lay, c) in enumerate(zip(layers, cnts)):
            b_indptr, b_src, b_dst, b_pos, b_eid, b_w, b_q, kept_nid, node_prob, cdev, t_indptr, t_edge = lay
            S, K, B = c.S, c.K, c.B
            blk = Block(self.g, K, S, b_indptr[:S + 1], b_src[:B], b_dst[:B], b_pos[:B], b_eid[:B], kept_nid[:K])
            blk._edge_weights, blk._q, blk._node_prob = b_w[:B], b_q[:B], node_prob[:K]
            blk._counts, blk._counts_dev = c, cdev
            if t_indptr is not None:
                blk._transposed = (t_indptr[:K + 1], t_edge[:max(B, 1)])
            ws = self.ws[n]
            blk._trace = dict(p=ws.p[:c.C], P=ws.P[:c.C], cand_nid=ws.cand_nid[:c.C], new_id=ws.new_id[:c.C])
            blocks.append(blk)
        return blocks
