"""CPU suite: the C-ABI library builds, loads, and exports every symbol include/bliss_gnn.h declares
(no compute calls here -- there is no GPU in the build container)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "bliss_gnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\*)\s+(bliss_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from bliss_gnn_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bliss_gnn.h but not exported"
        assert n in _lib.SIGNATURES or n in _lib.SPECIAL_SIGNATURES, f"{n} has no ctypes signature in bliss_gnn_amd/_lib.py"
    assert sorted(list(_lib.SIGNATURES) + list(_lib.SPECIAL_SIGNATURES)) == names


def test_counts_struct_layout():
    from bliss_gnn_amd import _lib
    assert ctypes.sizeof(_lib.LayerCounts) == 40 == _lib.lib.bliss_layer_counts_bytes()
    assert _lib.LayerCounts.c.offset == 32


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bliss_gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_flag_spin_bound_setter_is_host_only():
    """bliss_flag_set_spin_bound touches no device: a loop whose flags wait on collectives raises the bound before it captures
    its graphs (shard_static.PipelinedShardedTrainStep, worlds of more than one rank); invalid bounds are refused."""
    from bliss_gnn_amd import _lib
    assert _lib.lib.bliss_flag_set_spin_bound(1 << 27) == 0
    assert _lib.lib.bliss_flag_set_spin_bound(0) == _lib.EINVAL
    assert _lib.lib.bliss_flag_set_spin_bound(1 << 22) == 0     # (the default again)


def test_new_entry_points_refuse_bad_arguments_before_any_launch():
    """The round-3 entry points validate their arguments on the host (no kernel is launched for a refused call, so this runs
    without a GPU): null pointers, odd row lengths, misaligned scratch, a non-positive divisor."""
    import ctypes as C
    from bliss_gnn_amd import _lib
    lib, E = _lib.lib, _lib.EINVAL
    buf = (C.c_int64 * 64)()
    p = C.addressof(buf)
    assert lib.bliss_shard_place_rows(0, 8, p, p, 4, p, 8, 4, 8, 0) == E                  # no source
    assert lib.bliss_shard_place_rows(p, 8, p, p, 4, p, 8, 4, 7, 0) == E                  # odd row length
    assert lib.bliss_shard_take_rows(p, 0, 8, 4, 0, 0, 4, p, 8, 8, 0) == E                # no positions
    assert lib.bliss_shard_take_rows(p + 4, 1, 8, 4, p, 0, 4, p, 8, 8, 0) == E            # fp32 source not 8-byte aligned
    assert lib.bliss_shard_pack_rows(p, p, 4, 10, 10, p, 8, 8, p, 8, 0) == E              # empty node range
    assert lib.bliss_shard_zero_dense(0, 16, 0) == E
    assert lib.bliss_shard_candidates(p, 16, 0, p, p, p, p, p, 16, p + 4, p, 0) == E      # scratch not 8-byte aligned
    assert lib.bliss_cross_entropy_masked(p, 8, 0, 0, p, 4, p, 0, 4, p, 0.0, 3, p, p, 8, p, p, p, 0) == E   # divisor 0
    assert lib.bliss_cross_entropy_masked(p, 8, 0, 0, p, 4, 0, 0, 4, p, 8.0, 3, p, p, 8, p, p, p, 0) == E   # no label ids
    rows = (_lib.Exp3Block * 1)()
    assert lib.bliss_exp3_normalize_global_rows(rows, 1, 100, p, 96, 0) == E             # a row without buffers
    assert lib.bliss_exp3_normalize_global_rows(rows, 0, 100, p, 96, 0) == E
