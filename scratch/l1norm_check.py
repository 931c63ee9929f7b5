"""how often does torch's own bf16 L1 norm (F.normalize, bandit_sampler.py:249: fp32 accumulation in
torch's reduction order) land on a different bf16 value than the EXACT sum rounded once (what oracle and kernels use)?

Rows of |E_g| = 114,848,857 bf16 weights (the Reddit-like graph) evolved the way a long run evolves them: every step
multiplies ~180 K entries by bf16 factors exp(min(1, x)) (most of them 1.0, bandit_sampler.py:244-248) and divides the row
by its bf16 norm when that is not 1.0.  Per row-step: the exact norm (three-limb integer sum -> one rounding), torch's norm
with all threads, torch's norm with one thread.  CPU only (torch CPU is what the build container has; the reference's CUDA
reduction order is a third one).  Usage: python scratch/l1norm_check.py [steps] [E]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import numerics as nx

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
E = int(sys.argv[2]) if len(sys.argv) > 2 else 114848857
dev = torch.device(sys.argv[3]) if len(sys.argv) > 3 else torch.device("cpu")      # "cuda": torch-ROCm's own reduction kernel
gen = torch.Generator().manual_seed(0)
w = torch.ones(E, dtype=torch.bfloat16, device=dev)
nthreads = torch.get_num_threads()
dis_all = dis_one = dis_between = n_pass = 0
vals = []
t0 = time.time()
hot = torch.randint(0, E, (2_000_000,), generator=gen)              # the edges the sampler keeps coming back to
for step in range(steps):
    idx = hot[torch.randint(0, hot.numel(), (180_000,), generator=gen)].to(dev)
    x = torch.exp(torch.randn(180_000, generator=gen) * 2.0 - 4.0).clamp(max=1.0).bfloat16()       # delta_reward, capped at 1 (:244)
    f = torch.exp(x).to(dev)                                                                         # :246, bf16
    w[idx] = w[idx] * f                                                                              # :248 (last write wins on repeats, like torch)
    exact = nx.int_to_bf16(nx.row_exact_sum(w.cpu() if dev.type != "cpu" else w), nx.ROW_FRAC).to(dev)
    torch.set_num_threads(nthreads)
    t_all = w.norm(p=1, dim=0, keepdim=True)[0]
    torch.set_num_threads(1)
    t_one = w.norm(p=1, dim=0, keepdim=True)[0]
    torch.set_num_threads(nthreads)
    # (on "cuda" both calls run the same device kernel; the thread count is a CPU notion)
    dis_all += int(t_all.view(torch.int16) != exact.view(torch.int16))
    dis_one += int(t_one.view(torch.int16) != exact.view(torch.int16))
    dis_between += int(t_one.view(torch.int16) != t_all.view(torch.int16))
    vals.append((float(exact), float(t_all), float(t_one)))
    if float(exact) != 1.0:
        n_pass += 1
        w = w / exact.clamp_min(1e-12)
    if step % 10 == 0:
        print(step, round(time.time() - t0), vals[-1], "disagree all/one/between", dis_all, dis_one, dis_between, flush=True)
print(f"rows x steps: {steps}; renormalisation passes: {n_pass}; torch({nthreads} threads) != exact: {dis_all}; torch(1 thread) != exact: {dis_one}; "
      f"torch({nthreads}) != torch(1): {dis_between}")
print("norm values seen:", sorted(set(v[0] for v in vals)))
