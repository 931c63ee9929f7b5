"""A lean training loop that reproduces the reference's call sites around the hot path.

pytorch_lightning / dgl.dataloading.DataLoader are not available on this platform, so this module
replays, in order, exactly what they do to the sampler and the model each step
(train_lightning.py:100-168 training_step, :205-216 optimiser, :396-408 loader, :463-471 callback):

    seeds  = next batch of train ids (shuffled per epoch, drop_last)           DataLoader
    input_nodes, output_nodes, mfgs = sampler.sample(g, seeds)                 BlockSampler.sample
    x = mfgs[0].srcdata['features'];  y = mfgs[-1].dstdata['labels']          training_step :138-139
    loss = loss_fn(model(mfgs, x), y);  loss.backward();  optimiser.step()     :141-142 + Lightning
    sampler.exp3(mfgs, g)                                                      BatchSizeCallback :469-471
"""
import torch
import torch.nn as nn


class BatchLoader:
    """dgl.dataloading.DataLoader(g, train_nid, sampler, batch_size, shuffle=True, drop_last=True)
    reduced to its id stream (train_lightning.py:396-408).  Shuffles on the ids' device with its own
    generator so that the sampler's CPU random stream is untouched."""

    def __init__(self, ids, batch_size, shuffle=True, drop_last=True, seed=2):
        self.ids, self.bs, self.shuffle, self.drop_last = ids, int(batch_size), shuffle, drop_last
        self.gen = torch.Generator(device=ids.device)
        self.gen.manual_seed(seed)

    def __len__(self):
        n = self.ids.numel()
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        ids = self.ids
        if self.shuffle:
            ids = ids[torch.randperm(ids.numel(), generator=self.gen, device=ids.device)]
        for i in range(len(self)):
            yield ids[i * self.bs:(i + 1) * self.bs]

    def forever(self):
        while True:
            yield from iter(self)


class TrainStep:
    """One optimiser step of ModelLightning (train_lightning.py:50-216) with the bandit callback."""

    def __init__(self, g, sampler, model, lr=0.002, multilabel=False, bandit=True, grad_sync=None, exp3_sync=None):
        self.g, self.sampler, self.model = g, sampler, model
        self.loss_fn = nn.BCEWithLogitsLoss() if multilabel else nn.CrossEntropyLoss()   # :77-79
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)                           # :206
        self.bandit = bandit
        self.grad_sync, self.exp3_sync = grad_sync, exp3_sync
        self.num_steps = 0
        self.w = 0.99                                                                    # :76
        n_layers = len(sampler.nodes_per_layer)
        self.cum_sampled_nodes = [0.0] * (n_layers + 1)
        self.cum_sampled_edges = [0.0] * n_layers
        self.last = {}

    def _ema(self, mfgs):
        self.num_steps += 1                                                              # :103
        for i, mfg in enumerate(mfgs):                                                   # :104-110
            self.cum_sampled_nodes[i] = self.cum_sampled_nodes[i] * self.w + mfg.num_src_nodes()
            self.cum_sampled_edges[i] = self.cum_sampled_edges[i] * self.w + mfg.num_edges()
        i = len(mfgs)
        self.cum_sampled_nodes[i] = self.cum_sampled_nodes[i] * self.w + mfgs[-1].num_dst_nodes()   # :127-129

    def num_sampled_edges(self, i):                                                      # :91-98
        return self.cum_sampled_edges[i] * (1 - self.w) / (1 - self.w ** self.num_steps)

    def num_sampled_nodes(self, i):                                                      # :82-89
        return self.cum_sampled_nodes[i] * (1 - self.w) / (1 - self.w ** self.num_steps)

    def __call__(self, seeds):
        input_nodes, output_nodes, mfgs = self.sampler.sample(self.g, seeds)
        self._ema(mfgs)
        batch_inputs = mfgs[0].srcdata["features"]                                       # :138
        batch_labels = mfgs[-1].dstdata["labels"]                                        # :139
        batch_pred = self.model(mfgs, batch_inputs)                                      # :141
        loss = self.loss_fn(batch_pred, batch_labels)                                    # :142
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync(self.model)
        self.opt.step()
        if self.bandit:
            if self.exp3_sync is not None:
                self.exp3_sync(self.sampler, mfgs, self.g)
            else:
                self.sampler.exp3(mfgs, self.g)                                          # :469-471
        self.last = dict(loss=loss, mfgs=mfgs, pred=batch_pred, labels=batch_labels)
        return loss
