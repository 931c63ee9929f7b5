"""TEST DOUBLE: the local (per-shard) operations of bliss_gnn_amd.shard.ShardedPoissonBanditSampler restated with the
oracle's arithmetic on CPU tensors, so that the exchange logic (who sends what to whom, in which order the lists are
assembled) can run under gloo without a GPU.  The product path uses bliss_gnn_amd.shard._HipShardOps; this file is imported
by tests only."""
import numpy as np
import torch
import torch.distributed as dist

from oracle import bliss_oracle as bo
from oracle import numerics as nx


class _Blk:
    """Just enough of ShardBlock for the sampler and the comparisons."""

    def __init__(self, src, dst, pos, edge_weights, q, node_prob, src_nid, dst_pos, oblk, layer):
        self.src, self.dst, self.pos, self.dst_pos = src, dst, pos, dst_pos
        self._edge_weights, self._q, self._node_prob = edge_weights, q, node_prob
        self.srcdata, self.dstdata, self.edata = {"_ID": src_nid}, {"_ID": src_nid[dst_pos]}, {"_ID": pos}
        self.oblk, self._layer = oblk, layer

    def num_edges(self):
        return int(self.src.numel())

    def num_src_nodes(self):
        return int(self.srcdata["_ID"].numel())

    def num_dst_nodes(self):
        return int(self.dst_pos.numel())


class OracleShardOps:
    def __init__(self, shard, n_layers, eta, importance_sampling=True):
        self.shard, self.eta, self.imp = shard, eta, importance_sampling
        self.g = bo.CSC(shard.indptr, shard.indices, None)                    # local positions are the edge ids here
        self.w = torch.ones(n_layers, self.g.num_edges, dtype=torch.bfloat16)
        self.edge_w = bo.normalized_edata(self.g)
        self._st = {}

    def frontier_partials(self, n, layer, seeds_l):
        fr = bo.expand_frontier(self.g, seeds_l.long())
        q, _ = bo.exp3_edge_prob(self.g, fr, self.w[layer], self.eta)
        self._st[n] = (fr, q, layer)
        if seeds_l.numel() == 0:
            return torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int64)
        if self.imp:
            q_sum, _ = nx.exact_segment_sum(q, fr.dst_l, fr.n_seeds, nx.FRAC_DST)     # bandit_sampler.py:67
            t = (q / q_sum[fr.dst_l]) ** 2                                          # :71, :73
        else:
            t = torch.ones_like(q)
        fx = nx.bf16_to_fixed(t, nx.FRAC_SRC)
        C = fr.nid.numel()
        acc = torch.zeros(C, dtype=torch.int64).index_add_(0, fr.src_l, fx)
        touched = torch.zeros(C, dtype=torch.bool)
        touched[fr.src_l] = True
        touched[: fr.n_seeds] = True                                                # (the product sends its seeds' sums too)
        return fr.nid[touched].to(torch.int32), acc[touched]

    def importance(self, sums):
        if not self.imp:
            return (sums != 0).to(torch.bfloat16)
        return torch.sqrt(nx.fixed_to_bf16(sums, nx.FRAC_SRC))

    def scale(self, hist, n_cand, fanout, eps=0.9999):
        vals = nx.bits_to_bf16(torch.arange(hist.numel()))
        p_all = torch.repeat_interleave(vals, hist.long())
        assert p_all.numel() == n_cand
        _, c, iters = bo.poisson_scale(p_all, 0, fanout, eps)
        self._c, self._all_one, self._iters = c, n_cand <= fanout, iters

    def scale_result(self):
        return float(self._c), bool(self._all_one), int(self._iters)

    def keyed_select(self, ids, p, is_seed, seed, step, n):
        if self._all_one:
            P = torch.ones_like(p)
        else:
            pp = p.clone()
            pp[is_seed] = float("inf")                                              # bandit_sampler.py:403-404
            P = torch.minimum(pp * self._c, torch.ones_like(pp))                    # :406
        u = bo.keyed_uniform(seed, step, n, ids.long())
        return P, u < P.float()

    def build_block(self, n, kept_g, prob_g, seed_pos):
        fr, q, layer = self._st[n]
        K = kept_g.numel()
        V = self.g.num_nodes
        kidx = torch.full((V,), -1, dtype=torch.int64)
        kidx[kept_g.long()] = torch.arange(K)
        loc = kidx[fr.nid]                                                          # kept index of every LOCAL candidate
        chosen = torch.nonzero(loc >= 0).flatten()
        P_sg = torch.ones(fr.nid.numel(), dtype=torch.bfloat16)
        P_sg[chosen] = prob_g[loc[chosen]]
        ob = bo.generate_block(self.g, fr, chosen, P_sg, q, hajek=True)
        src = kidx[ob.src_nid[ob.src]].to(torch.int32)
        return _Blk(src, ob.dst.to(torch.int32), ob.eid.to(torch.int32), ob.edge_weights, ob.q_ij, prob_g, kept_g, seed_pos, ob, layer)

    def exp3_update(self, blk, embed_norm, delta_f):
        ob, layer = blk.oblk, blk._layer
        if ob.src.numel() == 0:
            return
        kidx = torch.full((self.g.num_nodes,), -1, dtype=torch.int64)
        kidx[blk.srcdata["_ID"].long()] = torch.arange(blk.num_src_nodes())
        en_local = embed_norm[kidx[ob.src_nid]]
        rewards = bo.exp3_rewards(ob, bo.sage_alpha(ob, self.edge_w), en_local)
        n_i = self.g.in_degrees()[ob.dst_nid].to(torch.int32).bfloat16()            # bandit_sampler.py:223
        d_r = (rewards / ob.node_prob[ob.src]) * (0.01 / n_i)[ob.dst]               # :240-242
        d_r = d_r.clone()
        d_r[d_r > 1] = 1                                                            # :244
        row = self.w[layer].clone()
        row[ob.eid] = row[ob.eid] * torch.exp(d_r)                                  # :246-248
        self.w[layer] = row
        blk.edata["rewards"] = rewards

    def normalize(self, idx, group=None):
        total = nx.row_exact_sum(self.w[idx])                                       # exact, scaled by 2^64
        limbs = torch.tensor([(total >> (32 * k)) & 0xFFFFFFFF for k in range(4)], dtype=torch.int64)
        dist.all_reduce(limbs, group=group)
        total = sum(int(limbs[k]) << (32 * k) for k in range(4))
        norm = nx.int_to_bf16(total, nx.ROW_FRAC)                                   # :249 the norm of the WHOLE row
        self.w[idx] = self.w[idx] / norm.clamp_min(1e-12)
