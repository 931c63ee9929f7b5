"""Host-side driver of the gfx950 sampler kernels (one layer = three C-ABI calls).

Shared by the four sampler classes.  Owns the dense per-node maps and the per-layer workspaces;
PyTorch is used for device memory and the current stream only.
"""
import ctypes as C

import torch

from . import _lib
from .graph import Block, Graph

_CHUNK = 1024


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


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
        self.local_id = torch.full((V,), -1, dtype=torch.int32, device=dev)
        self.first_pos = torch.full((V,), -1, dtype=torch.int32, device=dev)      # 0xFFFFFFFF
        self.acc_p2 = torch.zeros(V, dtype=torch.int64, device=dev)
        self.c_graph = _lib.Graph(g.indptr.data_ptr(), g.indices.data_ptr(), _ptr(g.eid), V, self.Eg)
        self.c_maps = _lib.NodeMaps(self.local_id.data_ptr(), self.first_pos.data_ptr(), self.acc_p2.data_ptr())
        self.chunk_cnt = torch.empty(max(self.Eg, V) // _CHUNK + 2, dtype=torch.int32, device=dev)
        self.counts_host = torch.empty(40, dtype=torch.uint8).pin_memory()

    def _read_counts(self, counts_dev):
        """One device->host sync: the sizes the reference obtains through .item()/boolean masks."""
        self.counts_host.copy_(counts_dev, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        c = _lib.LayerCounts.from_buffer_copy(self.counts_host.numpy().tobytes())
        if c.err:
            raise RuntimeError(f"sampler kernel error 0x{c.err:x}: {_lib.err_string(c.err)}")
        return c

    def sample_layer(self, w_pos, seeds, fanout, mode, eta, poisson=True, eps=0.9999, uniforms=None, chooser=None):
        """One iteration of the ``for block_id in reversed(range(L))`` loop of
        bandit_sampler.py:350-366 / ladies_sampler.py:112-122.

        ``uniforms``: optional fp32 CPU/GPU tensor replacing ``torch.rand(C)`` from the global CPU
        generator (the stream ATen's CPU ``torch.bernoulli`` consumes)."""
        g, dev = self.g, self.g.device
        st = _stream()
        S = int(seeds.numel())
        seeds = seeds.to(torch.int32).contiguous()
        V = self.V
        counts = torch.empty(40, dtype=torch.uint8, device=dev)
        seg_ptr = torch.empty(S + 1, dtype=torch.int32, device=dev)
        seed_acc = torch.empty(32 * S, dtype=torch.uint8, device=dev)
        cand_nid = torch.empty(V, dtype=torch.int32, device=dev)
        p = torch.empty(V, dtype=torch.bfloat16, device=dev)
        P = torch.empty(V, dtype=torch.bfloat16, device=dev)
        new_id = torch.empty(V, dtype=torch.int32, device=dev)
        kept_nid = torch.empty(V, dtype=torch.int32, device=dev)
        node_prob = torch.empty(V, dtype=torch.bfloat16, device=dev)
        ws = _lib.LayerWs(counts.data_ptr(), seg_ptr.data_ptr(), seed_acc.data_ptr(), self.chunk_cnt.data_ptr(),
                          cand_nid.data_ptr(), p.data_ptr(), P.data_ptr(), new_id.data_ptr(), kept_nid.data_ptr(),
                          node_prob.data_ptr(), V, V)
        eta_f = float(torch.tensor(eta, dtype=torch.float32))
        ome_f = float(torch.tensor(1.0 - eta, dtype=torch.float32))
        _lib.check(_lib.lib.bliss_frontier_prob(C.byref(self.c_graph), C.byref(self.c_maps), w_pos.data_ptr(),
                                                seeds.data_ptr(), S, mode, eta_f, ome_f, self.Eg, C.byref(ws), st),
                   "bliss_frontier_prob")
        c1 = self._read_counts(counts)                     # E, C  (the reference syncs here many times)
        if poisson:
            if uniforms is None:
                u = torch.rand(c1.C).pin_memory().to(dev, non_blocking=True)
            else:
                u = uniforms[: c1.C].to(dev, torch.float32).contiguous()
                assert u.numel() >= c1.C, "not enough uniforms supplied"
            _lib.check(_lib.lib.bliss_poisson_select(C.byref(ws), int(fanout), float(eps), u.data_ptr(), c1.C, st),
                       "bliss_poisson_select")
        else:
            raise NotImplementedError("multinomial (non-Poisson) selection lands with SURVEY 8f rank 4")
        cap_b = max(c1.E, 1)
        b_indptr = torch.empty(S + 1, dtype=torch.int32, device=dev)
        b_src = torch.empty(cap_b, dtype=torch.int32, device=dev)
        b_dst = torch.empty(cap_b, dtype=torch.int32, device=dev)
        b_pos = torch.empty(cap_b, dtype=torch.int32, device=dev)
        b_eid = torch.empty(cap_b, dtype=torch.int32, device=dev)
        b_w = torch.empty(cap_b, dtype=torch.bfloat16, device=dev)
        b_q = torch.empty(cap_b, dtype=torch.bfloat16, device=dev)
        out = _lib.BlockOut(b_indptr.data_ptr(), b_src.data_ptr(), b_dst.data_ptr(), b_pos.data_ptr(), b_eid.data_ptr(),
                            b_w.data_ptr(), b_q.data_ptr(), cap_b)
        _lib.check(_lib.lib.bliss_build_block(C.byref(self.c_graph), C.byref(self.c_maps), w_pos.data_ptr(),
                                              seeds.data_ptr(), S, mode, eta_f, ome_f, max(c1.E, 1), C.byref(ws),
                                              C.byref(out), st), "bliss_build_block")
        c2 = self._read_counts(counts)                     # K, B
        K, B = c2.K, c2.B
        blk = Block(g, K, S, b_indptr, b_src[:B], b_dst[:B], b_pos[:B], b_eid[:B], kept_nid[:K])
        blk._edge_weights = b_w[:B]
        blk._q = b_q[:B]
        blk._node_prob = node_prob[:K]
        blk._counts = c2
        blk._counts_dev = counts
        blk._trace = dict(p=p[: c2.C], P=P[: c2.C], cand_nid=cand_nid[: c2.C], new_id=new_id[: c2.C])
        return blk
