"""Bias gradient d.sum(0) of a tall bf16 matrix: aten::sum vs products with a ones vector (graph replay)."""
import torch, os
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
dev = torch.device("cuda:0")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n / 10
for R, N in ((5272, 256), (2224, 256), (256, 41)):
    d = torch.randn(R, N, device=dev).bfloat16()
    ones_r = torch.ones(1, R, device=dev, dtype=torch.bfloat16); ones_v = torch.ones(R, device=dev, dtype=torch.bfloat16)
    ref = d.float().sum(0)
    for name, f in (("d.sum(0)", lambda: d.sum(0)), ("ones[1,R] @ d", lambda: (ones_r @ d)[0]), ("mv(d.t(), ones)", lambda: torch.mv(d.t(), ones_v)),
                    ("d.t() @ ones[R,1]", lambda: (d.t() @ ones_v[:, None])[:, 0])):
        print("R=%5d N=%3d  %-20s %.1f us  max err %.3g" % (R, N, name, t(f), (f().float() - ref).abs().max()))
