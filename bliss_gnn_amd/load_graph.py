"""Dataset entry point with the reference's names (load_graph.py:69-80): ``load_dataset(name) -> (g, n_classes, multilabel)``.

The reference obtains every dataset through ``dgl.data`` / ``ogb``, which download on first use.  Neither package nor the
network exists on this platform, so this module reads the files those packages CACHE on disk, when a directory holding
them is given, and otherwise serves the seeded synthetic stand-ins of ``synth.CONFIGS``.  On-disk layouts
are [DGL-recalled] (no copy of the datasets is available to pin them against -- "parity unpinned" for the formats; the
round trip through files written in the same layout is what tests/test_load_graph.py checks):

  reddit   <root>/reddit/reddit_data.npz        feature [V,602] f32, label [V], node_types [V] (1 train, 2 val, 3 test)
           <root>/reddit/reddit_graph.npz       scipy.sparse.save_npz COO (row, col, data, shape)
  yelp / flickr (GraphSAINT layout)
           <root>/<name>/adj_full.npz           scipy.sparse.save_npz CSR (indptr, indices, data, shape)
           <root>/<name>/feats.npy              [V,F] float
           <root>/<name>/class_map.json         {node: class | [0/1]*C}
           <root>/<name>/role.json              {"tr": [...], "va": [...], "te": [...]}

Planetoid (cora/citeseer/pubmed) caches are Python pickles; they are not read (nothing here unpickles files).
The returned graph is the RAW dataset graph as an edge list; ``prep.prepare_graph`` turns it into the training CSC.
"""
import json
import os

import numpy as np
import torch


class EdgeListGraph:
    """What ``dgl.data.*[0]`` amounts to for this path: an edge list plus node frames."""

    def __init__(self, src, dst, num_nodes, ndata):
        self.src, self.dst, self._n, self.ndata = src, dst, int(num_nodes), ndata

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self.src.numel())

    def all_edges(self):
        return self.src, self.dst


def _masks(n, train, val, test):
    out = {}
    for key, ids in (("train_mask", train), ("val_mask", val), ("test_mask", test)):
        m = torch.zeros(n, dtype=torch.bool)
        m[torch.as_tensor(ids, dtype=torch.int64)] = True
        out[key] = m
    return out


def _coo_from_npz(path):
    z = np.load(path)                                                   # allow_pickle stays False
    fmt = z["format"].item() if "format" in z.files else b"coo"
    fmt = fmt.decode() if isinstance(fmt, bytes) else str(fmt)
    shape = tuple(int(x) for x in z["shape"])
    if fmt == "coo":
        return torch.from_numpy(z["row"].astype(np.int64)), torch.from_numpy(z["col"].astype(np.int64)), shape[0]
    if fmt == "csr":
        indptr, col = z["indptr"].astype(np.int64), z["indices"].astype(np.int64)
        row = np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(indptr))
        return torch.from_numpy(row), torch.from_numpy(col), shape[0]
    raise ValueError(f"unsupported sparse format {fmt!r} in {path}")


def load_reddit(root):
    """dgl.data.RedditDataset's cache.  Edge (row -> col) as DGL builds it with ``from_scipy``."""
    d = os.path.join(root, "reddit")
    z = np.load(os.path.join(d, "reddit_data.npz"))
    src, dst, n = _coo_from_npz(os.path.join(d, "reddit_graph.npz"))
    types = torch.from_numpy(z["node_types"].astype(np.int64))
    ndata = dict(features=torch.from_numpy(z["feature"]).bfloat16(),                        # load_graph.py:7
                 labels=torch.from_numpy(z["label"].astype(np.int64)),                      # load_graph.py:8
                 train_mask=types == 1, val_mask=types == 2, test_mask=types == 3)
    return EdgeListGraph(src, dst, n, ndata), int(ndata["labels"].max()) + 1


def load_graphsaint(root, name):
    """dgl.data.YelpDataset / FlickrDataset caches (GraphSAINT file set)."""
    d = os.path.join(root, name)
    src, dst, n = _coo_from_npz(os.path.join(d, "adj_full.npz"))
    feats = torch.from_numpy(np.load(os.path.join(d, "feats.npy"))).bfloat16()
    with open(os.path.join(d, "class_map.json")) as f:
        cmap = json.load(f)
    with open(os.path.join(d, "role.json")) as f:
        role = json.load(f)
    first = next(iter(cmap.values()))
    if isinstance(first, list):                                                             # multilabel (yelp)
        labels = torch.zeros(n, len(first), dtype=torch.int64)
        for k, v in cmap.items():
            labels[int(k)] = torch.tensor(v, dtype=torch.int64)
        n_classes = len(first)
    else:
        labels = torch.zeros(n, dtype=torch.int64)
        for k, v in cmap.items():
            labels[int(k)] = int(v)
        n_classes = int(labels.max()) + 1
    ndata = dict(features=feats, labels=labels, **_masks(n, role["tr"], role["va"], role["te"]))
    return EdgeListGraph(src, dst, n, ndata), n_classes


def toy():
    """ToyDataset, load_graph.py:92-120."""
    ndata = dict(features=torch.tensor([[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 0, 0, 0]], dtype=torch.float32),
                 labels=torch.tensor([0, 0, 1, 1, 1]), **_masks(5, range(5), [], []))
    return EdgeListGraph(torch.tensor([2, 3, 3, 4]), torch.tensor([0, 0, 1, 1]), 5, ndata)


def load_dataset(dataset_name, root=None):
    """load_graph.py:69-80.  ``root``: directory holding DGL's cached dataset folders (default ``$DGL_DOWNLOAD_DIR`` or
    ``~/.dgl``).  Raises FileNotFoundError (never downloads) when the files are not there."""
    root = root or os.environ.get("DGL_DOWNLOAD_DIR") or os.path.join(os.path.expanduser("~"), ".dgl")
    multilabel = False
    if dataset_name == "reddit":
        g, n_classes = load_reddit(root)
    elif dataset_name in ("yelp", "flickr"):
        g, n_classes = load_graphsaint(root, dataset_name)
        multilabel = dataset_name == "yelp"
        if multilabel:
            g.ndata["labels"] = g.ndata["labels"].to(torch.float32)                         # load_graph.py:74-75
    elif dataset_name in ("cora", "citeseer", "pubmed", "actor"):
        raise NotImplementedError(f"{dataset_name}: DGL caches it as Python pickles, which this build does not unpickle; "
                                  "use synth.CONFIGS for a graph of the same shape")
    elif dataset_name in ("ogbn-products", "ogbn-arxiv", "ogbn-papers100M"):
        raise NotImplementedError("ogb datasets need the ogb package (absent); out of scope (SURVEY.md 8f)")
    elif dataset_name == "toy":
        g, n_classes = toy(), 2
    else:
        raise ValueError("unknown dataset")                                                # load_graph.py:78
    return g, n_classes, multilabel
