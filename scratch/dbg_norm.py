import sys, torch
sys.path.insert(0, '.')
from bliss_gnn_amd import _lib
from oracle import numerics as nx
dev = torch.device('cuda:0')
gen = torch.Generator().manual_seed(0)
for n in (5, 100, 13199, 100003, 1 << 20):
    for off in (0, 1, 3, 7):
        vals = (torch.rand(n, generator=gen) * 3.0 / n).bfloat16()
        base = torch.zeros(n + 16, dtype=torch.bfloat16, device=dev)
        w = base[off:off + n]; w.copy_(vals)
        rs = torch.zeros(96, dtype=torch.int64, device=dev); sc = torch.zeros(98, dtype=torch.int64, device=dev)
        nrm = torch.zeros(1, dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib.bliss_row_sum(w.data_ptr(), n, rs.data_ptr(), 0), "row_sum")
        _lib.check(_lib.lib.bliss_exp3_normalize(w.data_ptr(), n, rs.data_ptr(), sc.data_ptr(), nrm.data_ptr(), 0), "normalize")
        torch.cuda.synchronize()
        exact = nx.row_exact_sum(vals)
        # bf16 norm of the exact sum
        import math
        normf = torch.tensor(exact / 2.0 ** 64, dtype=torch.float64)
        # reference: torch CPU division by the bf16-rounded norm
        want = (vals.float() / nrm.cpu().float()).bfloat16()
        got = w.cpu()
        ok = torch.equal(got.view(torch.int16), want.view(torch.int16))
        guard = bool((base[:off] == 0).all()) and bool((base[off + n:] == 0).all())
        fresh = torch.zeros(96, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib.bliss_row_sum(w.data_ptr(), n, fresh.data_ptr(), 0), "row_sum")
        tot = lambda r: sum(int(r[3 * s]) + (int(r[3 * s + 1]) << 32) + (int(r[3 * s + 2]) << 64) for s in range(32))
        print(n, off, "norm", float(nrm), "values ok", ok, "no stray writes", guard, "sum ok", tot(rs.cpu()) == tot(fresh.cpu()), "scratch clean", int(sc[1:].abs().sum()) == 0)
