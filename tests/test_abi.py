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
