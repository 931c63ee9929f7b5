"""Weight-gradient GEMMs of the input layer (dW = dZ^T X, reduction over ~11 K capacity rows): library call vs split-K via bmm."""
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
for R, N, K in ((11136, 256, 602), (5120, 256, 602), (5120, 256, 256), (2048, 41, 256)):
    dz = torch.randn(R, N, device=dev).bfloat16(); x = torch.randn(R, K, device=dev).bfloat16()
    ref = (dz.float().t() @ x.float())
    print("R=%d N=%d K=%d" % (R, N, K))
    out = dz.t() @ x
    print("   dz.t() @ x             %.1f us   max err %.3g" % (t(lambda: dz.t() @ x), (out.float() - ref).abs().max()))
    for S in (4, 8, 16, 32):
        if R % S: continue
        def f(S=S):
            p = torch.bmm(dz.view(S, R // S, N).transpose(1, 2), x.view(S, R // S, K))
            return p.sum(0, dtype=torch.float32).bfloat16()
        print("   bmm split %2d (bf16 partials) %.1f us   max err %.3g" % (S, t(f), (f().float() - ref).abs().max()))
        try:
            def g2(S=S):
                p = torch.bmm(dz.view(S, R // S, N).transpose(1, 2), x.view(S, R // S, K), out_dtype=torch.float32)
                return p.sum(0).bfloat16()
            print("   bmm split %2d (fp32 partials) %.1f us   max err %.3g" % (S, t(g2), (g2().float() - ref).abs().max()))
        except Exception as e:
            print("   out_dtype unsupported:", str(e)[:80])
