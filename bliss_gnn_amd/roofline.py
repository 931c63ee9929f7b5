"""Per-kernel roofline accounting for bench.py.

Durations come from HIP events that libbliss_gnn.so records around the selected kernel on the
stream it is launched on (bliss_prof_* in include/bliss_gnn.h).  ``achieved`` = algorithmic bytes
per launch / average launch duration; the algorithmic bytes of every kernel are the minimum HBM
traffic of that pass as a function of the layer's sizes, NOT what it moves.
"""
import ctypes as C

from . import _lib

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def kernel_names():
    n = _lib.lib.bliss_prof_kernel_count()
    return [_lib.lib.bliss_prof_kernel_name(i).decode() for i in range(n)]


def algorithmic_bytes(kernel, s, dims=None, layer=None):
    """Minimum bytes one launch must move for a layer with sizes s = {S,E,C,K,B} (ids int32, values bf16)."""
    S, E, Cn, K, B = s["S"], s["E"], s["C"], s["K"], s["B"]
    D = dims[layer] if dims is not None and layer is not None else 0
    table = {
        "k_frontier_pass1": 6 * E + 8 * S,                    # column indices + exp3 weights, per-seed pointers
        "k_frontier_pass2": 6 * E + 16 * S,                   # + per-seed sums in/out
        "k_frontier_pass3": 6 * E + 16 * S + 12 * Cn,         # + candidate ids out, per-candidate accumulators
        # binned candidate pipeline: the same minimum traffic as passes 1-3, split over its kernels (the (position, source,
        # term) records it writes and re-reads are implementation traffic, not algorithmic bytes)
        "k_col_sums": 2 * E + 24 * S,                         # exp3 weights of the seeds' columns, per-seed sums out
        "k_bin_scatter": 6 * E + 16 * S,                      # column indices + exp3 weights, per-seed sums in
        "k_bin_reduce": 12 * Cn,                              # per-candidate first position + sum out
        "k_cand_number": 14 * Cn,                             # candidate ids + p out
        "k_block_pass1": 6 * E + 16 * S + 6 * Cn,             # + new ids / P of the candidates
        "k_block_pass2": 6 * E + 16 * S + 6 * Cn + 20 * B,    # + the block's edges out (src,dst,pos,eid,w,q)
        "k_cand_finalize": 14 * Cn,
        "k_poisson_scale": 2 * Cn,
        "k_select_pass1": 8 * Cn, "k_select_pass2": 12 * Cn + 6 * K,
        "k_mt19937_uniform": 4 * Cn,
        "k_spmm_fwd": 4 * (S + 1) + 6 * B + 2 * K * D + 2 * S * D,
        "k_spmm_bwd": 4 * (K + 1) + 10 * B + 2 * S * D + 2 * K * D,
        "k_exp3_update": 12 * B + 4 * K + 4 * S,
        "k_embed_norm": 2 * K * D + 2 * K,
    }
    return table.get(kernel)


class KernelTimer:
    def __init__(self):
        self.names = kernel_names()

    def enable(self, which):
        """which: 'all', 'off' or a kernel name."""
        idx = {"all": -1, "off": -2}.get(which)
        if idx is None:
            idx = self.names.index(which)
        _lib.check(_lib.lib.bliss_prof_enable(idx), "bliss_prof_enable")
        _lib.check(_lib.lib.bliss_prof_reset(), "bliss_prof_reset")

    def read(self):
        out = {}
        for i, n in enumerate(self.names):
            ms, cnt = C.c_double(), C.c_int64()
            _lib.check(_lib.lib.bliss_prof_read(i, C.byref(ms), C.byref(cnt)), "bliss_prof_read")
            if cnt.value:
                out[n] = dict(total_ms=ms.value, launches=cnt.value, avg_us=1e3 * ms.value / cnt.value)
        return out
