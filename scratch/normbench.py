"""Time bliss_exp3_normalize on a row that needs the pass (norm != 1) and on one that does not."""
import sys, torch
sys.path.insert(0, '.')
from bliss_gnn_amd import _lib
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 113988362
for off in (0, 1):                                     # aligned and misaligned row start
    base = torch.full((n + 8,), 1.0 / n, dtype=torch.bfloat16, device=dev)
    w = base[off:off + n]
    rs = torch.zeros(96, dtype=torch.int64, device=dev); sc = torch.zeros(98, dtype=torch.int64, device=dev)
    nrm = torch.zeros(1, dtype=torch.bfloat16, device=dev)
    for rep in range(4):
        w.mul_(1.3 if rep % 2 == 0 else 1.0)           # odd reps: already normalised -> skipped
        _lib.check(_lib.lib.bliss_row_sum(w.data_ptr(), n, rs.data_ptr(), 0), "row_sum")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(_lib.lib.bliss_exp3_normalize(w.data_ptr(), n, rs.data_ptr(), sc.data_ptr(), nrm.data_ptr(), 0), "normalize")
        b.record(); torch.cuda.synchronize()
        fresh = torch.zeros(96, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib.bliss_row_sum(w.data_ptr(), n, fresh.data_ptr(), 0), "row_sum")
        tot = lambda r: sum(int(r[3 * s]) + (int(r[3 * s + 1]) << 32) + (int(r[3 * s + 2]) << 64) for s in range(32))
        print("offset %d rep %d: %.1f us  norm before %.6f  sum after %.6f  incremental == fresh: %s" % (
            off, rep, 1e3 * a.elapsed_time(b), float(nrm), w.float().sum().item(), tot(rs.cpu()) == tot(fresh.cpu())))
