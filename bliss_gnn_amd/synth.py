"""Seeded synthetic graphs shaped like the reference's datasets.

The reference downloads Cora/Pubmed/Reddit/Yelp through DGL (``load_graph.py:11-22``);
there is no network here, so benchmarks and tests use Chung-Lu graphs with the
datasets' |V|, |E|, feature width and class count.  Graph preparation follows
``train_lightning.py:334-342``: self loops removed then re-added LAST (so they carry the
highest edge ids), int32 ids, CSC only.
"""
import torch

# name -> (|V|, |E| before self loops, F, classes, train ids, batch, fanouts, multilabel)
CONFIGS = {
# features: "bow" = sparse non-negative rows normalised to sum 1 -- what the Planetoid datasets hold (Cora: binary bag of words,
#           ~18 of 1,433 words per paper; Pubmed: TF-IDF, ~50 of 500; DGL row-normalises both: recalled from the dataset
#           docs, no copy offline); "normal" = dense N(0,1) (Reddit: GloVe sums, Yelp: word2vec -- dense, unit scale).
#           The distinction matters to the BANDIT: with N(0,1) rows of width 1,433 every reward hits the cap
#           (bandit_sampler.py:244) and the reference's own weights leave bf16's range after a few hundred steps
#           (tests/golden/collapse0_normal_features: the reference run raises at step 616); with row-normalised rows
#           (norms ~0.2) the factors stay near 1 and thousand-step windows run, as they do on the real datasets.
    "cora":   dict(num_nodes=2708,   num_edges=10556,     feat=1433, classes=7,   n_train=140,    batch=32,  fanouts=[512, 256, 128],    multilabel=False, features="bow", nnz=18),
    "pubmed": dict(num_nodes=19717,  num_edges=88651,     feat=500,  classes=3,   n_train=60,     batch=32,  fanouts=[512, 256, 128],    multilabel=False, features="bow", nnz=50),
    "reddit": dict(num_nodes=232965, num_edges=114615892, feat=602,  classes=41,  n_train=153431, batch=256, fanouts=[4096, 2048, 1024], multilabel=False, features="normal"),
    "yelp":   dict(num_nodes=716847, num_edges=13954819,  feat=300,  classes=100, n_train=537635, batch=256, fanouts=[4096, 2048, 1024], multilabel=True, features="normal"),
}


def chung_lu_csc(num_nodes, num_edges, seed=0, device="cpu", sigma=1.0, chunk=1 << 26):
    """Directed Chung-Lu graph with lognormal(0, sigma) expected degrees, as CSC.

    Returns ``(indptr int64 [V+1], indices int32 [E], eid int32 [E])`` where ``eid`` is the
    COO edge id of every CSC position: non-loop edges are numbered dst-major in
    [0, E'), the V self loops take ids E'..E'+V-1 and sit LAST in their column."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    V = int(num_nodes)
    w = torch.exp(sigma * torch.randn(V, generator=gen, device=dev, dtype=torch.float32)).double()
    w = (w / w.sum()).float()
    keys = []
    left = int(num_edges)
    while left > 0:
        n = min(left, chunk)
        dst = torch.multinomial(w, n, replacement=True, generator=gen)
        src = torch.multinomial(w, n, replacement=True, generator=gen)
        m = src != dst
        keys.append(dst[m] * V + src[m])
        left -= n
    key = torch.unique(torch.cat(keys))            # sorted: dst-major, src ascending, no duplicates
    del keys
    dst = torch.div(key, V, rounding_mode="floor")
    src = key - dst * V
    Ep = key.numel()
    deg = torch.bincount(dst, minlength=V) + 1     # +1 self loop
    indptr = torch.zeros(V + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(deg, 0)
    E = Ep + V
    indices = torch.empty(E, dtype=torch.int32, device=dev)
    eid = torch.empty(E, dtype=torch.int32, device=dev)
    pos = torch.arange(Ep, device=dev) + dst       # dst earlier self loops precede edge i
    indices[pos] = src.to(torch.int32)
    eid[pos] = torch.arange(Ep, device=dev, dtype=torch.int32)
    loop_pos = indptr[1:] - 1
    indices[loop_pos] = torch.arange(V, device=dev, dtype=torch.int32)
    eid[loop_pos] = (Ep + torch.arange(V, device=dev)).to(torch.int32)
    return indptr, indices, eid


def node_data(num_nodes, feat, classes, n_train, seed=1, device="cpu", multilabel=False, features="normal", nnz=18):
    """features bf16 [V,F] (``load_graph.py:7``) -- ``features`` = "normal": N(0,1); "bow": about ``nnz`` random non-negative
    entries per row, rows normalised to sum 1 (the Planetoid datasets' row-normalised bags of words, see CONFIGS) -- labels,
    train ids (first n_train of a permutation)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    if features == "bow":
        dense = torch.rand(num_nodes, feat, generator=gen, device=dev, dtype=torch.float32)
        keep = dense < (float(nnz) / feat)
        keep[torch.arange(num_nodes, device=dev), torch.randint(0, feat, (num_nodes,), generator=gen, device=dev)] = True   # no empty row
        vals = torch.rand(num_nodes, feat, generator=gen, device=dev, dtype=torch.float32) + 0.5
        feats = torch.where(keep, vals, torch.zeros_like(vals))
        feats = (feats / feats.sum(1, keepdim=True)).bfloat16()
    else:
        feats = torch.randn(num_nodes, feat, generator=gen, device=dev, dtype=torch.float32).bfloat16()
    if multilabel:
        labels = (torch.rand(num_nodes, classes, generator=gen, device=dev) < 0.1).float()
    else:
        labels = torch.randint(0, classes, (num_nodes,), generator=gen, device=dev)
    perm = torch.randperm(num_nodes, generator=gen, device=dev)
    train_nid = perm[:n_train].to(torch.int32)
    return feats, labels, train_nid
