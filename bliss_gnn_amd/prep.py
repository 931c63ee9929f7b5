"""Graph preparation: the edge list a dataset yields -> the int32 CSC training graph.

Mirrors, in one call, what ``DataModule.__init__`` does to the loaded graph before any sampler sees it
(train_lightning.py:334-341, 373):

    g = dgl.remove_self_loop(g); g = dgl.add_self_loop(g)
    if undirected: src, dst = g.all_edges(); g.add_edges(dst, src)
    g = g.int(); ...; g = g.formats(["csc"])

and fixes the edge-id <-> CSC-position convention that the per-edge EXP3 weights depend on.  The work runs in
csrc/prep.hip (flag scan, scatter, one stable radix sort by destination, column starts by binary search).
"""
import torch

from . import _lib
from ._engine import _stream
from .graph import Graph


def prepare_graph(src, dst, num_nodes, undirected=False, ndata=None, edata=None):
    """``src``/``dst``: integer tensors [E] on the GPU (edge e goes src[e] -> dst[e]).  Returns a ``Graph`` whose
    ``eid`` gives the DGL edge id of every CSC position.  ``edata`` (optional, per INPUT edge) is not carried over:
    DGL's remove/add_self_loop renumber the edges and the reference never reads the datasets' edge features."""
    if not src.is_cuda:
        raise RuntimeError("prepare_graph runs on the GPU (csrc/prep.hip); move the edge list there first")
    if edata:
        raise NotImplementedError("edge features do not survive remove_self_loop/add_self_loop in the reference either")
    src, dst = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
    E, V, und = src.numel(), int(num_nodes), int(bool(undirected))
    assert dst.numel() == E
    cap = _lib.lib.bliss_graph_prepare_capacity(E, V, und)
    tmp_bytes = _lib.lib.bliss_graph_prepare_temp_bytes(E, V, und)
    if cap < 0 or tmp_bytes < 0:
        raise ValueError("graph too large for int32 ids (train_lightning.py:339-341 would keep int64)")
    dev = src.device
    indptr = torch.empty(V + 1, dtype=torch.int64, device=dev)
    indices = torch.empty(cap, dtype=torch.int32, device=dev)
    eid = torch.empty(cap, dtype=torch.int32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    temp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.bliss_graph_prepare(src.data_ptr(), dst.data_ptr(), E, V, und, indptr.data_ptr(), indices.data_ptr(),
                                            eid.data_ptr(), n_out.data_ptr(), err.data_ptr(), temp.data_ptr(), tmp_bytes,
                                            _stream()), "bliss_graph_prepare")
    n, bad = int(n_out.item()), int(err.item())
    if bad:
        raise ValueError("edge endpoint outside [0, num_nodes)")
    return Graph(indptr, indices[:n].clone(), eid[:n].clone(), ndata=ndata)
