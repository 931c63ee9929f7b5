"""CPU restatement of the BLISS-GNN hot path -- the parity ORACLE.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``bliss_gnn_amd``) never does and fails loudly without its HIP library.

Parity status: **unpinned w.r.t. DGL, pinned w.r.t. the reference's own Python.**
The reference has no tests, fixtures or golden vectors, and
its arithmetic lives in ``dgl==2.2.1`` (README.md:16), which is not installed
and cannot be.  What is pinned: ``tests/golden/make_golden.py`` runs the
reference's UNMODIFIED sampler classes (imported from /root/reference in the
build container) over a stand-in ``dgl`` module (``oracle/dgl_standin.py``)
that supplies only the graph primitives, and checks that this module's direct
restatement gives bit-identical results; those vectors are committed under
``tests/golden/``.  What stays unpinned: that the stand-in primitives match
DGL's (edge order of ``in_subgraph``, numbering of ``compact_graphs`` /
``to_block``, summation precision of ``copy_e_sum``) -- recalled, see
``oracle/numerics.py`` for the summation contract.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Graphs are plain CSC: ``indptr[|V|+1]``, ``indices[|E|]``
(source of each in-edge, grouped by destination) and an optional ``eid`` array
(edge id of each CSC position; identity when absent).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import numerics as nx

BF = torch.bfloat16


@dataclass
class CSC:
    indptr: torch.Tensor            # int64 [V+1]
    indices: torch.Tensor           # int32 [E]  source node of each in-edge
    eid: Optional[torch.Tensor] = None   # int32 [E] edge id per CSC position (None = identity)

    @property
    def num_nodes(self):
        return self.indptr.numel() - 1

    @property
    def num_edges(self):
        return self.indices.numel()

    def in_degrees(self):
        return (self.indptr[1:] - self.indptr[:-1])

    def edge_ids(self, pos):
        return pos if self.eid is None else self.eid[pos].to(torch.int64)


@dataclass
class Frontier:
    """``dgl.in_subgraph`` + ``dgl.compact_graphs`` (bandit_sampler.py:123-125)."""
    seg_ptr: torch.Tensor    # int64 [S+1] start of every seed's column in the frontier
    pos: torch.Tensor        # int64 [E] CSC position of every frontier edge
    eid: torch.Tensor        # int64 [E] edge id
    src_g: torch.Tensor      # int64 [E] global source id
    dst_l: torch.Tensor      # int64 [E] local destination id == seed index
    src_l: torch.Tensor      # int64 [E] local source id
    nid: torch.Tensor        # int64 [C] global id of every candidate (seeds first)
    n_seeds: int


@dataclass
class OBlock:
    """What ``generate_block`` returns (bandit_sampler.py:322-337), as arrays."""
    n_src: int
    n_dst: int
    indptr: torch.Tensor        # int64 [S+1] CSR over destination
    src: torch.Tensor           # int64 [B] local source id (block numbering)
    dst: torch.Tensor           # int64 [B] local destination id
    eid: torch.Tensor           # int64 [B] original edge id            edata[dgl.EID]
    edge_weights: torch.Tensor  # bf16  [B]                              edata['edge_weights']
    q_ij: Optional[torch.Tensor]       # bf16 [B]                        edata['q_ij'] (bandit only)
    node_prob: Optional[torch.Tensor]  # bf16 [K]                        srcdata['node_prob'] (bandit only)
    src_nid: torch.Tensor       # int64 [K]                              srcdata[dgl.NID]
    dst_nid: torch.Tensor       # int64 [S]                              dstdata[dgl.NID]
    trace: dict = field(default_factory=dict)   # intermediates for kernel-level tests


# --------------------------------------------------------------------------
# s1-s2  frontier expansion and candidate numbering
# --------------------------------------------------------------------------
def expand_frontier(g: CSC, seeds: torch.Tensor) -> Frontier:
    """bandit_sampler.py:123-125 / ladies_sampler.py:42-43.

    [DGL-recalled] ``in_subgraph`` lists the in-edges of the seeds grouped by seed
    in the given seed order and in CSC order inside a seed; ``compact_graphs(...,
    always_preserve=seeds)`` numbers the seeds 0..S-1 and then every other node by
    first appearance in that edge list, and keeps the edge order."""
    seeds = seeds.to(torch.int64)
    S = seeds.numel()
    start = g.indptr[seeds]
    deg = g.indptr[seeds + 1] - start
    seg_ptr = torch.zeros(S + 1, dtype=torch.int64)
    seg_ptr[1:] = torch.cumsum(deg, 0)
    E = int(seg_ptr[-1])
    dst_l = torch.repeat_interleave(torch.arange(S, dtype=torch.int64), deg)
    pos = start[dst_l] + (torch.arange(E, dtype=torch.int64) - seg_ptr[dst_l])
    src_g = g.indices[pos].to(torch.int64)
    # first-appearance numbering
    local = torch.full((g.num_nodes,), -1, dtype=torch.int64)
    local[seeds] = torch.arange(S, dtype=torch.int64)
    is_new = local[src_g] < 0
    new_src = src_g[is_new]
    if new_src.numel():
        uniq, first = np.unique(new_src.numpy(), return_index=True)
        order = np.argsort(first, kind="stable")
        new_nid = torch.from_numpy(uniq[order])
    else:
        new_nid = torch.zeros(0, dtype=torch.int64)
    local[new_nid] = S + torch.arange(new_nid.numel(), dtype=torch.int64)
    nid = torch.cat([seeds, new_nid])
    return Frontier(seg_ptr, pos, g.edge_ids(pos), src_g, dst_l, local[src_g], nid, S)


# --------------------------------------------------------------------------
# s3-s7  EXP3 edge probabilities
# --------------------------------------------------------------------------
def exp3_edge_prob(g: CSC, fr: Frontier, w_row: torch.Tensor, eta: float):
    """bandit_sampler.py:127-137.  ``w_row`` = exp3_weights[layer], bf16, by edge id."""
    S = fr.n_seeds
    w_e = w_row[fr.eid]                                           # :127
    w_sum, _ = nx.exact_segment_sum_rel(w_e, fr.dst_l, S, nx.FRAC_DST)  # :129 copy_e_sum
    w_div = w_e / w_sum[fr.dst_l]                                 # :131 e_div_v
    n_i = (fr.seg_ptr[1:] - fr.seg_ptr[:-1]).to(torch.int32)      # :133 g.in_degrees(seeds)
    a = (eta / n_i).bfloat16()                                    # :137 (self.eta / n_i).bfloat16()
    b = (1 - eta) * w_div                                         # :137
    q = a[fr.dst_l] + b                                           # :137 v_add_e
    return q, dict(w_e=w_e, w_sum=w_sum)


# --------------------------------------------------------------------------
# s8-s12  LADIES node importance
# --------------------------------------------------------------------------
def bandit_node_importance(fr: Frontier, q: torch.Tensor, importance_sampling=True):
    """BanditLadiesSampler.compute_prob, bandit_sampler.py:65-82."""
    C = fr.nid.numel()
    if importance_sampling:
        q_sum, _ = nx.exact_segment_sum(q, fr.dst_l, fr.n_seeds, nx.FRAC_DST)   # :67
        r = q / q_sum[fr.dst_l]                                              # :71 e_div_u on the reverse
        p2, _ = nx.exact_segment_sum(r ** 2, fr.src_l, C, nx.FRAC_SRC)        # :73
        return torch.sqrt(p2), dict(q_sum=q_sum, p2=p2)                       # :75
    prob = torch.ones(C).bfloat16()                                           # :79
    outdeg = torch.zeros(C, dtype=torch.int64).index_add_(0, fr.src_l, torch.ones_like(fr.src_l))
    prob[outdeg == 0] = 0                                                     # :81
    return prob, {}


def ladies_node_importance(fr: Frontier, w_e: torch.Tensor, importance_sampling=True):
    """LadiesSampler.compute_prob, ladies_sampler.py:44-51 (``w_e`` = g.edata['w'] on the frontier)."""
    C = fr.nid.numel()
    if importance_sampling:
        p2, _ = nx.exact_segment_sum(w_e ** 2, fr.src_l, C, nx.FRAC_SRC)       # :47
        return torch.sqrt(p2), dict(p2=p2)                                    # :48
    prob = torch.ones(C)                                                      # :50 (fp32 in the reference)
    outdeg = torch.zeros(C, dtype=torch.int64).index_add_(0, fr.src_l, torch.ones_like(fr.src_l))
    prob[outdeg == 0] = 0
    return prob, {}


# --------------------------------------------------------------------------
# s13-s16  Poisson scale: Sum_j min(c p_j, 1) ~= fanout
# --------------------------------------------------------------------------
def poisson_scale(prob: torch.Tensor, n_seeds: int, num: int, eps: float = 0.9999):
    """PoissonBanditLadiesSampler.compute_prob, bandit_sampler.py:391-406
    (identical text at ladies_sampler.py:150-164)."""
    one = torch.ones_like(prob)
    if prob.shape[0] <= num:                                                  # :392
        return one, 1.0, 0
    c = 1.0
    iters = 0
    for i in range(50):                                                       # :396
        S = torch.sum(torch.minimum(prob * c, one).to(torch.float64)).item()  # :397
        iters += 1
        if min(S, num) / max(S, num) >= eps:                                  # :398
            break
        c *= num / S                                                          # :401
    prob = prob.clone()
    prob[:n_seeds] = float("inf")                                             # :403-404 (seeds are local 0..S-1)
    return torch.minimum(prob * c, one), c, iters                             # :406


def poisson_draw(P: torch.Tensor, uniforms: Optional[torch.Tensor] = None):
    """select_neighbors, bandit_sampler.py:422-424.  With ``uniforms`` None this is
    the reference's own call and consumes the global CPU generator; otherwise the
    same comparison ``u < float(P)`` that ATen's serial CPU Bernoulli kernel makes
    (one 24-bit uniform per element, element order)."""
    if uniforms is None:
        return torch.arange(P.shape[0])[torch.bernoulli(P) == 1]
    return torch.arange(P.shape[0])[uniforms < P.float()]


def keyed_uniform(seed: int, step: int, layer: int, nid: torch.Tensor) -> torch.Tensor:
    """The counter-based uniforms of the SHARDED sampler (bliss_gnn_amd/shard.py, csrc/shard.hip:keyed_u24): no serial
    stream can be shared by destination-range shards (parity caveat), so the draw of candidate
    ``nid`` in sampling layer ``layer`` of step ``step`` is a pure function of (seed, step, layer, nid): the SplitMix64
    finaliser of a 64-bit key, top 24 bits, u = r * 2^-24 (the resolution of torch's CPU stream).  No reference
    counterpart (the reference is single-device); single-GPU sampling keeps torch's stream."""
    M = (1 << 64) - 1
    key = ((int(seed) * 0x9E3779B97F4A7C15) + int(step)) & M
    # finalise the (seed, step) key BEFORE the node id is mixed in: otherwise the step sits in the low bits beside the id and
    # an aligned block of 2^k ids sees the same 2^k uniforms, permuted, on consecutive steps (round-2 advice)
    key = ((key ^ (key >> 30)) * 0xBF58476D1CE4E5B9) & M
    key = ((key ^ (key >> 27)) * 0x94D049BB133111EB) & M
    key ^= key >> 31
    key ^= (int(layer) & 0xFF) << 56
    z = (np.uint64(key) ^ nid.numpy().astype(np.uint64))
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    r = (z >> np.uint64(40)).astype(np.float32)
    return torch.from_numpy(r * np.float32(1.0 / 16777216.0))


def multinomial_draw(prob: torch.Tensor, num: int, replace=False):
    """BanditLadiesSampler.select_neighbors, bandit_sampler.py:98 / ladies_sampler.py:68."""
    return torch.multinomial(prob, min(num, prob.shape[0]), replacement=replace)


# --------------------------------------------------------------------------
# s18-s25  block construction
# --------------------------------------------------------------------------
def generate_block(g: CSC, fr: Frontier, chosen: torch.Tensor, P_sg: torch.Tensor, W_sg: torch.Tensor,
                   hajek: bool) -> OBlock:
    """generate_block: bandit_sampler.py:285-337 (hajek=True) / ladies_sampler.py:81-107 (hajek=False)."""
    S = fr.n_seeds
    C = fr.nid.numel()
    chosen = chosen.to(torch.int64)
    seed_idx = torch.arange(S, dtype=torch.int64)                 # :285 find_indices_in(seeds, NID)
    u_nodes = torch.unique(torch.cat([chosen, seed_idx]))         # :287 union -> sorted
    in_u = torch.zeros(C, dtype=torch.bool); in_u[u_nodes] = True
    drawn = torch.zeros(C, dtype=torch.bool); drawn[chosen] = True
    # :289 node-induced subgraph (destinations are seeds, always inside), then :291-298 keep
    # only edges whose SOURCE was drawn
    keep = in_u[fr.src_l] & drawn[fr.src_l]
    new_id = torch.cumsum(in_u.to(torch.int64), 0) - 1            # sg numbering = rank inside u_nodes
    e_src = new_id[fr.src_l[keep]]
    e_dst = fr.dst_l[keep]                                        # seeds keep ids 0..S-1
    P = P_sg[u_nodes]                                             # :309
    W = W_sg[keep]                                                # :311
    W_tilde = W / P[e_src]                                        # :314 e_div_u
    K = u_nodes.numel()
    W_sum, _ = nx.exact_segment_sum(W_tilde.bfloat16(), e_dst, K, nx.FRAC_BLK)   # :316
    d = torch.zeros(K, dtype=torch.int64).index_add_(0, e_dst, torch.ones_like(e_dst)).to(torch.int32)  # :318
    if hajek:
        ratio = d / W_sum                                         # :320  int / bf16 -> bf16
        W_tilde = W_tilde * ratio[e_dst]                          # :320 e_mul_v
    else:
        W_tilde = W_tilde * (d / 1.0).bfloat16()[e_dst]           # ladies_sampler.py:97
    indptr = torch.zeros(S + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(d[:S].to(torch.int64), 0)
    # :322 to_block(sg, seeds): destinations first, then the remaining sources by first
    # appearance; identical to sg numbering, asserted here.
    first_seen = _first_appearance(e_src, S, K)
    assert torch.equal(first_seen, torch.arange(K, dtype=torch.int64)), "to_block order != subgraph order"
    nids = fr.nid[u_nodes]                                        # :306
    return OBlock(n_src=K, n_dst=S, indptr=indptr, src=e_src, dst=e_dst, eid=fr.eid[keep],
                  edge_weights=W_tilde.bfloat16() if W_tilde.dtype != BF else W_tilde,
                  q_ij=W if hajek else None, node_prob=P if hajek else None,
                  src_nid=nids, dst_nid=nids[:S],
                  trace=dict(u_nodes=u_nodes, keep=keep, W_sum=W_sum, d=d))


def _first_appearance(e_src, S, K):
    """src numbering ``to_block`` would produce: 0..S-1, then first appearance among edges."""
    seen = torch.zeros(K, dtype=torch.bool); seen[:S] = True
    order = list(range(S))
    src_np = e_src.numpy()
    rest = src_np[src_np >= S]
    if rest.size:
        uniq, first = np.unique(rest, return_index=True)
        order += uniq[np.argsort(first, kind="stable")].tolist()
    # nodes without any edge (cannot happen for sampled non-seeds) keep their rank
    missing = sorted(set(range(K)) - set(order))
    order += missing
    return torch.tensor(order, dtype=torch.int64)


# --------------------------------------------------------------------------
# a11  sample_blocks
# --------------------------------------------------------------------------
def sample_blocks_bandit(g: CSC, seeds: torch.Tensor, fanouts, exp3_weights: torch.Tensor, eta: float,
                         poisson: bool = True, importance_sampling: bool = True,
                         uniforms: Optional[List[torch.Tensor]] = None, uniform_fn=None):
    """(Poisson)BanditLadiesSampler.sample_blocks, bandit_sampler.py:341-367.

    ``exp3_weights`` bf16 [L, |E|] by edge id.  ``uniforms`` (optional) = one fp32
    vector per layer in sampling order (last layer first) replacing the generator;
    ``uniform_fn(n, cand_nid) -> fp32 [C]`` (optional) = keyed mode: the uniform of a candidate is a
    function of the sampling layer n and its global node id (see keyed_uniform)."""
    blocks = []
    seed_nodes = seeds.to(torch.int64)
    for n, block_id in enumerate(reversed(range(len(fanouts)))):          # :350
        num = fanouts[block_id]
        fr = expand_frontier(g, seed_nodes)
        q, tr1 = exp3_edge_prob(g, fr, exp3_weights[block_id], eta)       # :354
        p, tr2 = bandit_node_importance(fr, q, importance_sampling)        # :356 (base class part)
        if poisson:
            P, c, iters = poisson_scale(p, fr.n_seeds, num)
            u = None if uniforms is None else uniforms[n][: P.shape[0]]
            if uniform_fn is not None:
                u = uniform_fn(n, fr.nid)
            chosen = poisson_draw(P, u)                                   # :360
        else:
            P, c, iters = p, 1.0, 0
            chosen = multinomial_draw(p, num)
        blk = generate_block(g, fr, chosen, P, q, hajek=True)             # :362
        blk.trace.update(tr1); blk.trace.update(tr2)
        blk.trace.update(p=p, P=P, c=c, iters=iters, q=q, cand_nid=fr.nid, E=fr.pos.numel(),
                         src_l=fr.src_l, dst_l=fr.dst_l, chosen=chosen)
        seed_nodes = blk.src_nid                                          # :364
        blocks.insert(0, blk)                                             # :366
    return seed_nodes, seeds, blocks


def sample_blocks_ladies(g: CSC, seeds: torch.Tensor, fanouts, edge_w: torch.Tensor, poisson: bool = True,
                         importance_sampling: bool = True, uniforms: Optional[List[torch.Tensor]] = None):
    """(Poisson)LadiesSampler.sample_blocks, ladies_sampler.py:109-123; ``edge_w`` = g.edata['w'] by edge id."""
    blocks = []
    seed_nodes = seeds.to(torch.int64)
    for n, block_id in enumerate(reversed(range(len(fanouts)))):
        num = fanouts[block_id]
        fr = expand_frontier(g, seed_nodes)
        w_e = edge_w[fr.eid]                                              # :46 / :120
        p, tr = ladies_node_importance(fr, w_e, importance_sampling)
        if poisson:
            P, c, iters = poisson_scale(p, fr.n_seeds, num)
            u = None if uniforms is None else uniforms[n][: P.shape[0]]
            chosen = poisson_draw(P, u)
        else:
            P, c, iters = p, 1.0, 0
            chosen = multinomial_draw(p, num)
        blk = generate_block(g, fr, chosen, P, w_e, hajek=False)
        blk.trace.update(tr)
        blk.trace.update(p=p, P=P, c=c, iters=iters, cand_nid=fr.nid, E=fr.pos.numel(), chosen=chosen)
        seed_nodes = blk.src_nid
        blocks.insert(0, blk)
    return seed_nodes, seeds, blocks


# --------------------------------------------------------------------------
# a3  normalized_edata
# --------------------------------------------------------------------------
def normalized_edata(g: CSC) -> torch.Tensor:
    """bandit_sampler.py:20-27 with weight=None: w_e = 1 / indeg(dst(e)), bf16, by edge id."""
    deg = g.in_degrees()
    ones = torch.ones(g.num_edges, dtype=BF)
    dst = torch.repeat_interleave(torch.arange(g.num_nodes, dtype=torch.int64), deg)
    v, _ = nx.exact_segment_sum(ones, dst, g.num_nodes, nx.FRAC_DST)       # :25 update_all(copy_e, sum)
    w_pos = (1 / v[dst]) * ones                                           # :26-27
    if g.eid is None:
        return w_pos
    out = torch.empty_like(w_pos)
    out[g.eid.to(torch.int64)] = w_pos
    return out


# --------------------------------------------------------------------------
# a13-a16  EXP3 reward + weight update
# --------------------------------------------------------------------------
def sage_alpha(blk: OBlock, edge_w: torch.Tensor):
    """calculate_alpha, model != 'gat': bandit_sampler.py:157 (block inherits edata['w'])."""
    return edge_w[blk.eid]


def gat_alpha(blk: OBlock, a_ij: torch.Tensor):
    """calculate_alpha, model == 'gat': bandit_sampler.py:148-154."""
    S = blk.n_dst
    q_sum, _ = nx.exact_segment_sum(blk.q_ij, blk.dst, S, nx.FRAC_DST)        # :150
    a_sum, _ = nx.exact_segment_sum(a_ij, blk.dst, S, nx.FRAC_DST)            # :151
    frac = torch.nan_to_num(a_ij / a_sum[blk.dst])                         # :152-153
    return frac * q_sum[blk.dst]                                           # :154 e_dot_v on scalars == product


def exp3_rewards(blk: OBlock, alpha: torch.Tensor, embed_norm: torch.Tensor):
    """calculate_rewards, bandit_sampler.py:180-193."""
    k_i = (blk.indptr[1:] - blk.indptr[:-1]).to(torch.int32).bfloat16()    # :180
    a = torch.nan_to_num((alpha ** 2) / k_i[blk.dst], posinf=0)           # :186-187
    h = (embed_norm ** 2)[blk.src] / (blk.q_ij ** 2)                      # :189 u_div_e
    return a * h                                                          # :191


def exp3_update_row(g: CSC, blk: OBlock, w_row: torch.Tensor, rewards: torch.Tensor, delta: float = 0.01):
    """update_exp3_weights, bandit_sampler.py:221-249.  Returns the new row and the bf16 norm."""
    n_i = g.in_degrees()[blk.dst_nid].to(torch.int32).bfloat16()          # :223
    r_hat = rewards / blk.node_prob[blk.src]                              # :240 e_div_u
    d_r = r_hat * (delta / n_i)[blk.dst]                                  # :242 e_mul_v
    d_r = d_r.clone()
    d_r[d_r > 1] = 1                                                      # :244
    ex = torch.exp(d_r)                                                   # :246
    w_row = w_row.clone()
    w_row[blk.eid] = w_row[blk.eid] * ex                                  # :248
    norm = nx.int_to_bf16(nx.row_exact_sum(w_row), nx.ROW_FRAC)           # :249 ||w||_1, exact then bf16
    denom = norm.clamp_min(1e-12)                                         # F.normalize eps
    return w_row / denom, norm, dict(exp_rewards=ex, delta_reward=d_r)


def exp3(g: CSC, blocks: List[OBlock], exp3_weights: torch.Tensor, edge_w: torch.Tensor,
         embed_norms: List[torch.Tensor], a_ij: Optional[List[torch.Tensor]] = None):
    """exp3, bandit_sampler.py:251-267 (SAGE/GCN alpha unless ``a_ij`` is given)."""
    out = exp3_weights.clone()
    traces = []
    for idx, blk in enumerate(blocks):
        alpha = sage_alpha(blk, edge_w) if a_ij is None else gat_alpha(blk, a_ij[idx])
        rewards = exp3_rewards(blk, alpha, embed_norms[idx])
        out[idx], norm, tr = exp3_update_row(g, blk, out[idx], rewards)
        tr.update(rewards=rewards, norm=norm, alpha=alpha)
        traces.append(tr)
    return out, traces


# --------------------------------------------------------------------------
# a17-a18  SAGE forward (fp32 reference of the floating-point kernels)
# --------------------------------------------------------------------------
def embed_norm_ref(h: torch.Tensor):
    """model.py:318-320: ||h_j||_2 per source row, fp32 math, returned in fp32."""
    return torch.linalg.vector_norm(h.float(), dim=1)


def spmm_mean_ref(blk: OBlock, h: torch.Tensor, edge_weight: Optional[torch.Tensor]):
    """[DGL-recalled] SAGEConv 'mean': update_all(u_mul_e('h','_edge_weight'), mean) -- fp32."""
    S = blk.n_dst
    msg = h.float()[blk.src]
    if edge_weight is not None:
        msg = msg * edge_weight.float()[:, None]
    out = torch.zeros(S, h.shape[1], dtype=torch.float32).index_add_(0, blk.dst, msg)
    deg = (blk.indptr[1:] - blk.indptr[:-1]).clamp(min=1).float()
    return out / deg[:, None]


def sage_conv_ref(blk: OBlock, h, W_self, b_self, W_neigh, edge_weight):
    """[DGL-recalled] dglnn.SAGEConv(in,out,'mean').forward(block, h, edge_weight) in fp32.
    ``W_*`` are [out,in] like nn.Linear; fc_neigh is applied BEFORE aggregation iff in > out."""
    h = h.float()
    in_f, out_f = W_neigh.shape[1], W_neigh.shape[0]
    h_self = h[: blk.n_dst]
    if in_f > out_f:
        neigh = spmm_mean_ref(blk, h @ W_neigh.float().t(), edge_weight)
    else:
        neigh = spmm_mean_ref(blk, h, edge_weight) @ W_neigh.float().t()
    return h_self @ W_self.float().t() + b_self.float() + neigh


# --------------------------------------------------------------------------
# a19  GATv2 forward in the reference's own arithmetic (bf16 tensors, torch CPU ops)
# --------------------------------------------------------------------------
def edge_softmax_ref(blk: OBlock, e: torch.Tensor):
    """[DGL-recalled] dglnn.functional.edge_softmax (model.py:89) = four ops in the dtype of ``e`` [B, H]: per-destination
    max, exp(e - max), per-destination sum (exact sum of the rounded terms, one rounding: the copy_e_sum contract of
    numerics.py), division."""
    S, H = blk.n_dst, e.shape[1]
    mx = torch.full((S, H), -float("inf")).scatter_reduce(0, blk.dst[:, None].expand(-1, H), e.float(), "amax").to(e.dtype)
    score = torch.exp(e - mx[blk.dst])
    ssum = torch.stack([nx.exact_segment_sum_rel(score[:, h].contiguous(), blk.dst, S, nx.FRAC_DST)[0] for h in range(H)], 1)
    return score / ssum[blk.dst]


def gatv2_conv(blk: OBlock, h: torch.Tensor, p: dict, slope: float):
    """custom_GATv2Conv.forward, model.py:63-112, as the reference builds it (share_weights=True, bias=False, :152-153):
    ``p`` = dict(fc_src [H*D, in], attn [1,H,D], res_fc [H*D, in] or None, res_kind 0 none / 1 Linear / 2 Identity, H, D,
    act: F.elu or nothing).  Returns (rst [S,H,D], e [B,H] -- the PRE-softmax logits the reference hands back as
    "attention", :108-110)."""
    H, D, S = p["H"], p["D"], blk.n_dst
    feat_src = torch.nn.functional.linear(h, p["fc_src"]).view(-1, H, D)                   # :70
    feat_dst = feat_src[:S]                                                                # :72, :78
    e = torch.nn.functional.leaky_relu(feat_src[blk.src] + feat_dst[blk.dst], slope)       # :82-85 u_add_v, leaky_relu
    e = (e * p["attn"]).sum(dim=-1)                                                        # :86   [B, H]
    a = edge_softmax_ref(blk, e)                                                           # :88-90
    # :98 update_all(u_mul_e('el','a','m'), sum('m','ft')): fp32 products and sums in edge order, one rounding
    # ([DGL-recalled]: the message tensor is not materialised; see dgl_standin.Graph.update_all)
    m = feat_src.float()[blk.src] * a.float()[:, :, None]
    rst = torch.zeros(S, H, D, dtype=torch.float32).index_add_(0, blk.dst, m).to(h.dtype)  # :99
    if p["res_kind"] == 1:
        rst = rst + torch.nn.functional.linear(h[:S], p["res_fc"]).view(S, -1, D)          # :101-103
    elif p["res_kind"] == 2:
        rst = rst + h[:S].view(S, -1, D)
    if p["act"]:
        rst = torch.nn.functional.elu(rst)                                                 # :105-106 (train_lightning.py:587)
    return rst, e


def gatv2_forward(blocks: List[OBlock], x: torch.Tensor, params: List[dict], slope: float = 0.2):
    """GATv2.forward, model.py:207-234 (no dropout): per layer embed_norm (:211-213), the layer, a_ij = mean of the logits
    over the heads (:224-227), flatten / mean over the heads (:228-232).  Returns (logits [S, classes], per-layer traces)."""
    h = x.bfloat16()                                                                       # :208
    traces = []
    for l, (blk, p) in enumerate(zip(blocks, params)):
        en = torch.reshape(torch.norm(h, dim=1, keepdim=True), (-1,))                      # :211-213
        rst, e = gatv2_conv(blk, h, p, slope)                                              # :214-223
        a_ij = torch.mean(e, dim=1)                                                        # :224-227
        traces.append(dict(embed_norm=en, rst=rst, e=e, a_ij=a_ij, h_in=h))
        h = rst.flatten(1) if l < len(blocks) - 1 else rst.mean(1)                         # :228-232
    return h, traces


def prepare_graph(src, dst, num_nodes, undirected=False):
    """train_lightning.py:334-341, 373 restated: remove_self_loop, add_self_loop, optional add_edges(dst, src), then
    CSC by a STABLE sort on destination ([DGL-recalled] columns keep ascending edge id, so the self loop is last).
    Pinned by the reference's ToyDataset (load_graph.py:96) -> tests/test_oracle.py."""
    src, dst = torch.as_tensor(src).long(), torch.as_tensor(dst).long()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    loops = torch.arange(num_nodes)
    src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
    if undirected:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    order = torch.sort(dst, stable=True).indices
    indptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.bincount(dst, minlength=num_nodes), 0)
    return CSC(indptr, src[order].to(torch.int32), order.to(torch.int32))
