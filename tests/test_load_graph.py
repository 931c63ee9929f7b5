"""Dataset readers: files written in the cached layouts DGL uses are read back to the same graph.
The layouts themselves are recalled, not pinned (no dataset copy exists offline) -- see bliss_gnn_amd/load_graph.py."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch


def _lg():
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location(
        "bliss_load_graph", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bliss_gnn_amd", "load_graph.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bliss_load_graph"] = mod
    spec.loader.exec_module(mod)            # standalone: the readers must not need the HIP library
    return mod


def test_reddit_layout(tmp_path):
    lg = _lg()
    rng = np.random.default_rng(0)
    V, E = 50, 400
    row, col = rng.integers(0, V, E), rng.integers(0, V, E)
    os.makedirs(tmp_path / "reddit")
    sp.save_npz(tmp_path / "reddit" / "reddit_graph.npz", sp.coo_matrix((np.ones(E), (row, col)), shape=(V, V)))
    feat = rng.standard_normal((V, 6)).astype(np.float32)
    label, types = rng.integers(0, 4, V), rng.integers(1, 4, V)
    np.savez(tmp_path / "reddit" / "reddit_data.npz", feature=feat, label=label, node_types=types, node_ids=np.arange(V))
    g, n_classes, multilabel = lg.load_dataset("reddit", str(tmp_path))
    assert (g.num_nodes(), g.num_edges(), multilabel) == (V, E, False) and n_classes == int(label.max()) + 1
    assert np.array_equal(g.src.numpy(), row) and np.array_equal(g.dst.numpy(), col)
    assert torch.equal(g.ndata["features"], torch.from_numpy(feat).bfloat16()) and g.ndata["labels"].dtype == torch.int64
    assert np.array_equal(g.ndata["train_mask"].numpy(), types == 1) and np.array_equal(g.ndata["test_mask"].numpy(), types == 3)


@pytest.mark.parametrize("name", ["yelp", "flickr"])
def test_graphsaint_layout(tmp_path, name):
    lg = _lg()
    rng = np.random.default_rng(1)
    V, C = 40, 5
    adj = sp.random(V, V, density=0.1, format="csr", random_state=2)
    os.makedirs(tmp_path / name)
    sp.save_npz(tmp_path / name / "adj_full.npz", adj)
    feats = rng.standard_normal((V, 7))
    np.save(tmp_path / name / "feats.npy", feats)
    if name == "yelp":
        cmap = {str(i): rng.integers(0, 2, C).tolist() for i in range(V)}
    else:
        cmap = {str(i): int(rng.integers(0, C)) for i in range(V)}
        cmap["0"] = C - 1
    perm = rng.permutation(V).tolist()
    role = dict(tr=perm[:20], va=perm[20:30], te=perm[30:])
    json.dump(cmap, open(tmp_path / name / "class_map.json", "w"))
    json.dump(role, open(tmp_path / name / "role.json", "w"))
    g, n_classes, multilabel = lg.load_dataset(name, str(tmp_path))
    coo = adj.tocoo()
    assert g.num_nodes() == V and n_classes == C and multilabel == (name == "yelp")
    assert np.array_equal(g.src.numpy(), coo.row) and np.array_equal(g.dst.numpy(), coo.col)
    assert g.ndata["labels"].dtype == (torch.float32 if name == "yelp" else torch.int64)
    assert g.ndata["labels"].shape == ((V, C) if name == "yelp" else (V,))
    assert sorted(torch.nonzero(g.ndata["val_mask"]).flatten().tolist()) == sorted(role["va"])
    assert g.ndata["features"].dtype == torch.bfloat16


def test_toy_and_errors(tmp_path):
    lg = _lg()
    g, n_classes, multilabel = lg.load_dataset("toy")
    assert (g.num_nodes(), g.num_edges(), n_classes, multilabel) == (5, 4, 2, False)        # load_graph.py:96
    with pytest.raises(ValueError):
        lg.load_dataset("nope")
    with pytest.raises(FileNotFoundError):
        lg.load_dataset("reddit", str(tmp_path))                                          # never downloads
    with pytest.raises(NotImplementedError):
        lg.load_dataset("cora", str(tmp_path))
