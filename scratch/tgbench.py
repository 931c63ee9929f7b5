"""Time the MFMA tile GEMM (csrc/sage.hip) on the three layer shapes of the Reddit-like step vs the library path."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bliss_gnn_amd import nn as bnn
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def timeit(fn, n=20):
    """device time per call: n calls recorded into one HIP graph (no host launch overhead), replayed"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); gr.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * n) * 1e3
V, F = 232965, 602
table = torch.randn(V, F, generator=g).bfloat16().to(dev)
for name, K, S, Kin, N, gather in (("L0 pair (gather)", 11100, 4900, 602, 256, True), ("L0 pair exact", 7300, 3300, 602, 256, True),
                                   ("L2 pair", 1900, 256, 256, 41, False)):
    ids = torch.randint(0, V, (K,), generator=g).to(torch.int32).to(dev)
    x = table if gather else torch.randn(K, Kin, generator=g).bfloat16().to(dev)
    wn = torch.randn(N, Kin, generator=g).bfloat16().to(dev); ws = torch.randn(N, Kin, generator=g).bfloat16().to(dev); b = torch.randn(N).bfloat16().to(dev)
    z = torch.empty(K, N, dtype=torch.bfloat16, device=dev); y = torch.empty(S, N, dtype=torch.bfloat16, device=dev)
    rows = torch.empty(K, Kin, dtype=torch.bfloat16, device=dev); nrm = torch.empty(K, dtype=torch.bfloat16, device=dev)
    a1 = bnn._tg_args(x, wn, z, K, ids=ids if gather else None, a_copy=rows if gather else None, in_norm=nrm)
    a2 = bnn._tg_args(x, ws, y, S, ids=ids if gather else None, bias=b)
    t = timeit(lambda: bnn._tile_gemm(a1, a2))
    def lib():
        xx = table[ids.long()] if gather else x
        torch.nn.functional.linear(xx, wn); torch.nn.functional.linear(xx[:S], ws, b)
    print(f"{name:20s} tile_gemm {t:7.1f} us   library (gather + 2 linear) {timeit(lib):7.1f} us")
S, D = 1900, 256
agg = torch.randn(S, D, generator=g).bfloat16().to(dev); h = torch.randn(4900, D, generator=g).bfloat16().to(dev)
wn = torch.randn(D, D, generator=g).bfloat16().to(dev); ws = torch.randn(D, D, generator=g).bfloat16().to(dev); b = torch.randn(D).bfloat16().to(dev)
out = torch.empty(S, D, dtype=torch.bfloat16, device=dev); nrm = torch.empty(S, dtype=torch.bfloat16, device=dev)
a = bnn._tg_args(agg, wn, out, S, a2=h[:S], w2=ws, bias=b, out_norm=nrm, relu=True)
print(f"{'L1 dual':20s} tile_gemm {timeit(lambda: bnn._tile_gemm(a)):7.1f} us   library {timeit(lambda: torch.relu(torch.nn.functional.linear(agg, wn) + torch.nn.functional.linear(h[:S], ws, b))):7.1f} us")
