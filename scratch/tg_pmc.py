"""A few plain launches of the L0-shaped tile GEMM for a rocprofv3 --pmc pass."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bliss_gnn_amd import nn as bnn
dev = torch.device("cuda"); g = torch.Generator().manual_seed(0)
V, F, K, S, N = 232965, 602, 7300, 3300, 256
table = torch.randn(V, F, generator=g).bfloat16().to(dev)
ids = torch.randint(0, V, (K,), generator=g).to(torch.int32).to(dev)
wn = torch.randn(N, F, generator=g).bfloat16().to(dev); ws = torch.randn(N, F, generator=g).bfloat16().to(dev); b = torch.randn(N).bfloat16().to(dev)
z = torch.empty(K, N, dtype=torch.bfloat16, device=dev); y = torch.empty(S, N, dtype=torch.bfloat16, device=dev)
rows = torch.empty(K, F, dtype=torch.bfloat16, device=dev); nrm = torch.empty(K, dtype=torch.bfloat16, device=dev)
a1 = bnn._tg_args(table, wn, z, K, ids=ids, a_copy=rows, in_norm=nrm); a2 = bnn._tg_args(table, ws, y, S, ids=ids, bias=b)
for _ in range(10):
    bnn._tile_gemm(a1, a2)
torch.cuda.synchronize()
