"""CPU port of one whole training step (sample -> SAGE fwd/bwd -> Adam -> exp3) built on the oracle.

TEST / BASELINE INFRASTRUCTURE ONLY: used by ``bench.py``'s ``cpu_baseline`` leg (kind "port": DGL-CPU
cannot be produced here) and by tests.  Plain torch-CPU ops; the model is the
[DGL-recalled] SAGEConv('mean') of model.py:303-333 with index_add_ message passing.
"""
import torch
import torch.nn as nn

from . import bliss_oracle as bo


class RefSAGEConv(nn.Module):
    def __init__(self, in_f, out_f):
        super().__init__()
        self.in_f, self.out_f = in_f, out_f
        self.fc_neigh = nn.Linear(in_f, out_f, bias=False)
        self.fc_self = nn.Linear(in_f, out_f, bias=True)
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def _agg(self, blk, h, w):
        msg = h[blk.src] * w.to(h.dtype)[:, None]
        out = torch.zeros(blk.n_dst, h.shape[1], dtype=h.dtype).index_add_(0, blk.dst, msg)
        deg = (blk.indptr[1:] - blk.indptr[:-1]).clamp(min=1).to(h.dtype)
        return out / deg[:, None]

    def forward(self, blk, h, w):
        if self.in_f > self.out_f:
            neigh = self._agg(blk, self.fc_neigh(h), w)
        else:
            neigh = self.fc_neigh(self._agg(blk, h, w))
        return self.fc_self(h[: blk.n_dst]) + neigh


class RefSAGE(nn.Module):
    def __init__(self, in_f, hid, n_cls, n_layers, dropout):
        super().__init__()
        dims = [in_f] + [hid] * (n_layers - 1) + [n_cls]
        self.layers = nn.ModuleList(RefSAGEConv(dims[i], dims[i + 1]) for i in range(n_layers))
        self.dropout = nn.Dropout(dropout)

    def forward(self, blocks, x):
        h, norms = x, []
        for l, (layer, blk) in enumerate(zip(self.layers, blocks)):
            norms.append(torch.linalg.vector_norm(h.detach(), dim=1))          # model.py:318-320
            h = layer(blk, h, blk.edge_weights)
            if l < len(self.layers) - 1:
                h = self.dropout(torch.relu(h))
        return h, norms


class RefTrainStep:
    def __init__(self, g: bo.CSC, feats, labels, fanouts, eta, in_f, hid, n_cls, lr=0.002, dropout=0.1, multilabel=False):
        self.g, self.feats, self.labels, self.fanouts, self.eta = g, feats, labels, fanouts, eta
        self.model = RefSAGE(in_f, hid, n_cls, len(fanouts), dropout).bfloat16()
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.loss_fn = nn.BCEWithLogitsLoss() if multilabel else nn.CrossEntropyLoss()
        self.w = torch.ones(len(fanouts), g.num_edges, dtype=torch.bfloat16)
        self.edge_w = bo.normalized_edata(g)

    def __call__(self, seeds):
        inp, _, blocks = bo.sample_blocks_bandit(self.g, seeds, self.fanouts, self.w, self.eta)
        x, y = self.feats[inp], self.labels[seeds.long()]
        pred, norms = self.model(blocks, x)
        loss = self.loss_fn(pred, y)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        self.w, _ = bo.exp3(self.g, blocks, self.w, self.edge_w, [n.bfloat16() for n in norms])
        return loss, blocks
