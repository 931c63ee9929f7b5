import torch
dev = torch.device("cuda:0")
E = 113988365
w = (torch.rand(E, device=dev) * 1e-8 + 4e-9).bfloat16()
y = torch.empty_like(w)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for name, f in (("copy y<-w", lambda: y.copy_(w)), ("in place w*=1", lambda: w.mul_(1.0)), ("read only sum", lambda: w.sum(dtype=torch.float32)),
                ("fill", lambda: y.zero_())):
    us = t(f)
    byts = {"copy y<-w": 4 * E, "in place w*=1": 4 * E, "read only sum": 2 * E, "fill": 2 * E}[name]
    print("%-16s %.1f us  %.2f TB/s" % (name, us, byts / us / 1e6))
