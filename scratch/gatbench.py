"""Per-kernel timings of the GATv2 message passing on the blocks of a Reddit-like step: the fused per-destination kernels
(csrc/gat_fused.hip) with and without attention dropout against the separate kernels (csrc/gat.hip).  usage: gatbench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bliss_gnn_amd as bg
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc
from bliss_gnn_amd.nn import GATv2Conv, _GatFusedMP, _GatLogits, _EdgeSoftmax, _GatAggregate

dev = torch.device("cuda:0")
cfg = CONFIGS["reddit"]
ip, ix, ei = chung_lu_csc(cfg["num_nodes"], cfg["num_edges"], seed=0, device=dev)
g = bg.Graph(ip, ix, ei)
g.edata["w"] = bg.normalized_edata(g)
s = bg.PoissonBanditLadiesSampler(cfg["fanouts"], eta=0.1, model="gat")
torch.manual_seed(0)
seeds = torch.randperm(cfg["num_nodes"], device=dev)[:256].to(torch.int32)
_, _, blocks = s.sample_blocks(g, seeds)
H, D = 4, 256

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(ev[:-1], ev[1:]))
    return t[len(t) // 2]

for li, blk in enumerate(blocks[:2]):
    K, S, B = blk.num_src_nodes(), blk.num_dst_nodes(), blk.num_edges()
    deg = blk.in_degrees()
    print(f"layer {li}: K {K} S {S} B {B} max in-degree {int(deg.max())} mean {B / S:.1f}")
    feat = (torch.randn(K, H * D, device=dev) * 0.3).bfloat16().requires_grad_(True)
    attn = (torch.randn(1, H, D, device=dev) * 0.2).bfloat16().requires_grad_(True)
    layer = GATv2Conv(8, D, H, 0.0, 0.1, 0.2, False, None, bias=False, share_weights=True, allow_zero_in_degree=True).to(dev).bfloat16()
    st = layer._fused_state(dev)
    gout = torch.randn(S, H * D, device=dev).bfloat16()
    for p in (0.0, 0.1):
        def fwd():
            with torch.no_grad():
                return _GatFusedMP.apply(feat.detach(), attn.detach(), blk, H, D, 0.2, p, st)
        print(f"  fused forward  p={p}: {timeit(fwd):8.1f} us")
        def fb():
            feat.grad = attn.grad = None
            rst, e = _GatFusedMP.apply(feat, attn, blk, H, D, 0.2, p, st)
            rst.backward(gout)
        print(f"  fused fwd+bwd  p={p}: {timeit(fb):8.1f} us")
    # where a workgroup's time goes: the kernels' phase stamps (100 MHz), one launch each
    from bliss_gnn_amd import _lib
    seg = _lib.lib.bliss_gat_segment_edges()
    cap = S + B // seg + 8
    st_f = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    st_b = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib.bliss_gat_fused_stamps(st_f.data_ptr(), st_b.data_ptr()), "stamps")
    feat.grad = attn.grad = None
    rst, e = _GatFusedMP.apply(feat, attn, blk, H, D, 0.2, 0.0, st)
    rst.backward(gout)
    torch.cuda.synchronize()
    _lib.check(_lib.lib.bliss_gat_fused_stamps(None, None), "stamps")
    for what, stamps, names in (("forward", st_f, ["row resolved", "pass 1 (gathers + logits)", "barrier", "max exchange", "pass 2 (softmax)",
                                                     "pass 3 (aggregate)", "reduce + store"]),
                                ("backward by destination", st_b, ["row resolved", "pass 1 (gathers + d a)", "barrier", "t exchange",
                                                                     "pass 2 (d e)", "pass 3 (d er, d attn)", "reduce + store"])):
        t = stamps.view(cap, 8).cpu().double()
        t = t[t[:, 7] > 0]
        d = (t[:, 1:] - t[:, :-1]) / 100
        print(f"  {what}: {t.shape[0]} workgroups, launch span {(t[:, 7].max() - t[:, 0].min()) / 100:.1f} us, workgroup life mean "
              f"{d.sum(1).mean():.2f} us, max {d.sum(1).max():.1f} us")
        print("    " + "; ".join(f"{n} {d[:, i].mean():.2f}" for i, n in enumerate(names)))
    def ufwd():
        with torch.no_grad():
            e = _GatLogits.apply(feat.detach(), attn.detach(), blk, H, D, 0.2)
            a = _EdgeSoftmax.apply(e, blk, H)
            return _GatAggregate.apply(a, feat.detach(), blk, H, D)
    print(f"  separate forward    : {timeit(ufwd):8.1f} us")
    def ufb():
        feat.grad = attn.grad = None
        e = _GatLogits.apply(feat, attn, blk, H, D, 0.2)
        a = _EdgeSoftmax.apply(e, blk, H)
        _GatAggregate.apply(a, feat, blk, H, D).backward(gout)
    print(f"  separate fwd+bwd    : {timeit(ufb):8.1f} us")
