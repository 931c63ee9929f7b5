"""LADIES samplers on MI355X -- same names and call sites as the reference's ``ladies_sampler.py``
(``train_lightning.py:358-360``): ``(PoissonLadiesSampler | LadiesSampler)(fanouts)``, static edge
weights ``g.edata['w']`` instead of the EXP3 state.  Kernels: include/bliss_gnn.h, BLISS_MODE_LADIES.
"""
import torch

from . import _lib
from ._engine import LayerEngine
from .bandit_sampler import BlockSampler, find_indices_in, normalized_edata, union  # noqa: F401  (same helpers, ladies_sampler.py:6-22)
from .graph import NID


class LadiesSampler(BlockSampler):
    """ladies_sampler.py:24-123.  ``select_neighbors`` (:54-69) = ``torch.multinomial`` on the device-computed importances."""

    _poisson = False

    def __init__(self, nodes_per_layer, importance_sampling=True, weight="w", out_weight="edge_weights",
                 replace=False, allow_zero_in_degree=False):
        super().__init__()
        self.nodes_per_layer = nodes_per_layer
        self.importance_sampling = importance_sampling
        self.edge_weight = weight
        self.output_weight = out_weight
        self.replace = replace
        self.allow_zero_in_degree = allow_zero_in_degree
        self.eps = 0.9999
        self._engine = None

    def _mode(self):
        # (the reference's non-importance branch builds fp32 ones, ladies_sampler.py:50: 0 / 1 are the same numbers in bf16 on
        # the device; the host draw of the multinomial variant gets them as fp32, see sample_blocks)
        return _lib.MODE_LADIES | (0 if self.importance_sampling else _lib.MODE_UNIFORM_NODES)

    def sample_blocks(self, g, seed_nodes, exclude_eids=None, uniforms=None):
        """ladies_sampler.py:109-123."""
        g = self._graph(g)
        self._bind(g)
        w_pos = g.edata_by_position(self.edge_weight)                    # :114
        output_nodes = seed_nodes
        order = list(reversed(range(len(self.nodes_per_layer))))         # :112
        fan = [self.nodes_per_layer[b] for b in order]
        if self._poisson:
            blks = self._engine.sample_blocks([w_pos] * len(order), seed_nodes, fan, self._mode(), 0.0, self.eps, uniforms)
        else:                                                            # select_neighbors :54-69 (torch.multinomial)
            blks = self._engine.sample_blocks_multinomial([w_pos] * len(order), seed_nodes, fan, self._mode(), 0.0, self.replace,
                                                          fp32_importance=not self.importance_sampling)
        blocks = []
        for blk in blks:
            blk.edata[self.output_weight] = blk._edge_weights            # :100
            blocks.insert(0, blk)
        return blocks[0].srcdata[NID], output_nodes, blocks              # :121,:123


    # -- static-shape variant (graph-capturable; Poisson only): same contract as PoissonBanditLadiesSampler's ------------
    def _graph(self, g):
        from .graph import as_graph
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        return as_graph(g, self._graphs)

    def _bind(self, g):
        g = self._graph(g)
        if self._engine is None or self._engine.g is not g:
            self._engine = LayerEngine(g)
        return self._engine

    def sample_blocks_static(self, g, seed_nodes, slot=0, chain_rng=False, external_rng=False, part=None, last_block=True, ready_flag=0):
        if not self._poisson:
            raise NotImplementedError("the multinomial draw is torch.multinomial on the host: no static-shape variant")
        g = self._graph(g)
        eng = self._bind(g)
        w_pos = g.edata_by_position(self.edge_weight)
        order = list(reversed(range(len(self.nodes_per_layer))))
        blks = eng.enqueue_static([w_pos] * len(order), seed_nodes, [self.nodes_per_layer[b] for b in order], self._mode(), 0.0,
                                  self.eps, slot=slot, chain_rng=chain_rng, external_rng=external_rng, part=part, last_block=last_block, ready_flag=ready_flag)
        blocks = []
        for blk in blks:
            blk.edata[self.output_weight] = blk._edge_weights
            blocks.insert(0, blk)
        return blocks[0].srcdata[NID], seed_nodes, blocks

    def finish_static(self, slot=0, commit=True):
        return self._engine.finish(slot, commit)

    def check_errors(self):
        pass                                               # no bandit state; sampler errors surface through finish()


class PoissonLadiesSampler(LadiesSampler):
    """ladies_sampler.py:125-183."""

    _poisson = True

    def __init__(self, nodes_per_layer, importance_sampling=True, weight="w", out_weight="edge_weights",
                 allow_zero_in_degree=False):
        super().__init__(nodes_per_layer, importance_sampling, weight, out_weight, False, allow_zero_in_degree)
