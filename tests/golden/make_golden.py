#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's unmodified ``bandit_sampler.py`` / ``ladies_sampler.py`` are imported
from /root/reference (nothing is copied) on top of ``oracle/dgl_standin.py`` -- the only
part that is not the reference's own code, because dgl==2.2.1 cannot be installed here.
Every case is also pushed through ``oracle/bliss_oracle.py`` and the script aborts unless
the two agree bit for bit; what is written is the REFERENCE run's output.

bf16 tensors are stored as their uint16 bit patterns (numpy has no bfloat16).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import dgl_standin
from oracle import bliss_oracle as bo
from bliss_gnn_amd.synth import chung_lu_csc

dgl = dgl_standin.install()
sys.path.insert(0, "/root/reference")
import bandit_sampler as ref_bandit      # noqa: E402  (the reference, unmodified)
import ladies_sampler as ref_ladies      # noqa: E402
import model as ref_model                # noqa: E402  (custom_GATv2Conv / GATv2 / SAGE, model.py, unmodified)

NID = dgl.NID


def bits(x):
    return (x.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF).numpy().astype(np.uint16)


def ref_graph(indptr, indices, eid):
    g = dgl_standin.graph_from_csc(indptr, indices, eid)
    g.edata["w"] = ref_bandit.normalized_edata(g)          # train_lightning.py:362
    return g


def block_arrays(prefix, b, out, bandit):
    src, dst = b.edges()
    out[prefix + "src"] = src.numpy().astype(np.int64)
    out[prefix + "dst"] = dst.numpy().astype(np.int64)
    out[prefix + "eid"] = b.edata[dgl.EID].numpy().astype(np.int64)
    out[prefix + "edge_weights"] = bits(b.edata["edge_weights"].bfloat16())
    out[prefix + "src_nid"] = b.srcdata[NID].numpy().astype(np.int64)
    out[prefix + "dst_nid"] = b.dstdata[NID].numpy().astype(np.int64)
    if bandit:
        out[prefix + "q_ij"] = bits(b.edata["q_ij"])
        out[prefix + "node_prob"] = bits(b.srcdata["node_prob"])


def check_block(ob, b, bandit, what):
    src, dst = b.edges()
    assert torch.equal(ob.src, src.long()), what + " src"
    assert torch.equal(ob.dst, dst.long()), what + " dst"
    assert torch.equal(ob.eid, b.edata[dgl.EID].long()), what + " eid"
    assert torch.equal(ob.src_nid, b.srcdata[NID].long()), what + " src_nid"
    assert torch.equal(ob.dst_nid, b.dstdata[NID].long()), what + " dst_nid"
    assert np.array_equal(bits(ob.edge_weights), bits(b.edata["edge_weights"].bfloat16())), what + " edge_weights"
    if bandit:
        assert np.array_equal(bits(ob.q_ij), bits(b.edata["q_ij"])), what + " q_ij"
        assert np.array_equal(bits(ob.node_prob), bits(b.srcdata["node_prob"])), what + " node_prob"


def bandit_case(name, indptr, indices, eid, seeds_per_step, fanouts, eta, torch_seed, poisson=True, importance_sampling=1):
    """Several consecutive train steps: sample_blocks -> (synthetic embed_norm) -> exp3."""
    g = ref_graph(indptr, indices, eid)
    og = bo.CSC(indptr, indices, eid)
    cls = ref_bandit.PoissonBanditLadiesSampler if poisson else ref_bandit.BanditLadiesSampler
    sampler = cls(fanouts, importance_sampling=importance_sampling, node_embedding="features", num_steps=1000, eta=eta, model="sage")
    o_w = torch.ones(len(fanouts), og.num_edges, dtype=torch.bfloat16)
    edge_w = bo.normalized_edata(og)
    assert np.array_equal(bits(edge_w), bits(g.edata["w"])), "normalized_edata"
    out = dict(indptr=indptr.numpy(), indices=indices.numpy(), eid=eid.numpy(), fanouts=np.array(fanouts),
               eta=np.array(eta), torch_seed=np.array(torch_seed), n_steps=np.array(len(seeds_per_step)),
               edge_w=bits(edge_w), poisson=np.array(int(poisson)), importance_sampling=np.array(int(importance_sampling)))
    gen = torch.Generator().manual_seed(1234 + torch_seed)
    for step, seeds in enumerate(seeds_per_step):
        out[f"s{step}_seeds"] = seeds.numpy()
        torch.manual_seed(torch_seed + step)
        inp, outp, mfgs = sampler.sample_blocks(g, seeds)
        torch.manual_seed(torch_seed + step)
        o_inp, o_outp, o_blocks = bo.sample_blocks_bandit(og, seeds, fanouts, o_w, eta, poisson=poisson,
                                                           importance_sampling=bool(importance_sampling))
        assert torch.equal(o_inp, inp.long()), "input_nodes"
        embed = []
        for l, (b, ob) in enumerate(zip(mfgs, o_blocks)):
            check_block(ob, b, True, f"{name} step{step} layer{l}")
            block_arrays(f"s{step}_l{l}_", b, out, True)
            out[f"s{step}_l{l}_c"] = np.array(ob.trace["c"])
            out[f"s{step}_l{l}_E"] = np.array(ob.trace["E"])
            out[f"s{step}_l{l}_cand_nid"] = ob.trace["cand_nid"].numpy()
            out[f"s{step}_l{l}_p"] = bits(ob.trace["p"])
            out[f"s{step}_l{l}_P"] = bits(ob.trace["P"])
            en = (torch.rand(b.num_src_nodes(), generator=gen) * 30).bfloat16()   # stands in for ||h_j|| (model.py:318)
            b.srcdata["embed_norm"] = en
            embed.append(en)
            out[f"s{step}_l{l}_embed_norm"] = bits(en)
        sampler.exp3(mfgs, g)                                   # train_lightning.py:471
        o_w, traces = bo.exp3(og, o_blocks, o_w, edge_w, embed)
        for l, (b, tr) in enumerate(zip(mfgs, traces)):
            assert np.array_equal(bits(tr["rewards"]), bits(b.edata["rewards"])), f"{name} rewards {step}/{l}"
            out[f"s{step}_l{l}_rewards"] = bits(b.edata["rewards"])
        same = np.array_equal(bits(o_w), bits(sampler.exp3_weights))
        if not same:
            d = (bits(o_w) != bits(sampler.exp3_weights)).sum()
            raise SystemExit(f"{name}: exp3_weights differ from the reference after step {step} in {d} entries")
        out[f"s{step}_exp3_weights"] = bits(sampler.exp3_weights)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in list(out.items())[:0]})


def ladies_case(name, indptr, indices, eid, seeds, fanouts, torch_seed, poisson=True, importance_sampling=True):
    g = ref_graph(indptr, indices, eid)
    og = bo.CSC(indptr, indices, eid)
    cls = ref_ladies.PoissonLadiesSampler if poisson else ref_ladies.LadiesSampler
    sampler = cls(fanouts, importance_sampling=importance_sampling)
    edge_w = bo.normalized_edata(og)
    out = dict(indptr=indptr.numpy(), indices=indices.numpy(), eid=eid.numpy(), fanouts=np.array(fanouts),
               torch_seed=np.array(torch_seed), seeds=seeds.numpy(), edge_w=bits(edge_w), poisson=np.array(int(poisson)),
               importance_sampling=np.array(int(importance_sampling)))
    torch.manual_seed(torch_seed)
    inp, outp, mfgs = sampler.sample_blocks(g, seeds)
    torch.manual_seed(torch_seed)
    o_inp, _, o_blocks = bo.sample_blocks_ladies(og, seeds, fanouts, edge_w, poisson=poisson, importance_sampling=importance_sampling)
    assert torch.equal(o_inp, inp.long())
    for l, (b, ob) in enumerate(zip(mfgs, o_blocks)):
        check_block(ob, b, False, f"{name} layer{l}")
        block_arrays(f"l{l}_", b, out, False)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def _hook_layers(layers, rec):
    """Record every layer's input rows and raw output (forward hooks; the reference's modules are not modified)."""
    hs = []
    for l, layer in enumerate(layers):
        hs.append(layer.register_forward_pre_hook(lambda m, a, l=l: rec.__setitem__(f"l{l}_in", a[1].detach().clone())))
        hs.append(layer.register_forward_hook(lambda m, a, o, l=l: rec.__setitem__(f"l{l}_out", o)))
    return hs


def gat_case(name, V, E, F, hidden, classes, heads, fanouts, batch, eta, residual, n_steps, seed):
    """The reference's OWN model.GATv2 (custom_GATv2Conv.forward, model.py:48-112, 207-234) and exp3 with model='gat'
    (bandit_sampler.py:146-154, 251-267) over blocks drawn by its PoissonBanditLadiesSampler: bf16 parameters and
    activations on the CPU, F.elu, no dropout (train_lightning.py:581-596 minus the stochastic masks).  Every step the
    direct restatement (oracle.gatv2_forward / oracle.exp3 with a_ij) must agree bit for bit before anything is written."""
    import torch.nn.functional as Fn
    ip, ix, ei = chung_lu_csc(V, E, seed=seed)
    g = ref_graph(ip, ix, ei)
    og = bo.CSC(ip, ix, ei)
    feats = (torch.randn(V, F, generator=torch.Generator().manual_seed(seed + 1)) * 0.5).bfloat16()
    g.ndata["features"] = feats
    edge_w = bo.normalized_edata(og)
    sampler = ref_bandit.PoissonBanditLadiesSampler(fanouts, importance_sampling=1, node_embedding="features", num_steps=1000,
                                                    eta=eta, model="gat")
    torch.manual_seed(seed + 2)
    net = ref_model.GATv2(len(fanouts), F, hidden, classes, heads, Fn.elu, 0.0, 0.0, 0.2, residual).bfloat16()
    out = dict(indptr=ip.numpy(), indices=ix.numpy(), eid=ei.numpy(), fanouts=np.array(fanouts), eta=np.array(eta),
               torch_seed=np.array(seed), n_steps=np.array(n_steps), features=bits(feats), heads=np.array(heads),
               hidden=np.array(hidden), classes=np.array(classes), residual=np.array(int(residual)), edge_w=bits(edge_w))
    params = []
    for l, layer in enumerate(net.gatv2_layers):
        out[f"p{l}_fc_src"] = bits(layer.fc_src.weight.detach())
        out[f"p{l}_attn"] = bits(layer.attn.detach().reshape(-1))
        rw = layer.res_fc.weight.detach() if isinstance(layer.res_fc, torch.nn.Linear) else None
        if rw is not None:
            out[f"p{l}_res_fc"] = bits(rw)
        out[f"p{l}_res_kind"] = np.array(0 if layer.res_fc is None else (1 if rw is not None else 2))   # none / Linear / Identity
        params.append(dict(fc_src=layer.fc_src.weight.detach(), attn=layer.attn.detach(), res_fc=rw,
                           res_kind=int(out[f"p{l}_res_kind"]), H=heads[l], D=layer._out_feats, act=layer.activation is not None))
    o_w = torch.ones(len(fanouts), og.num_edges, dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(seed + 3)
    for step in range(n_steps):
        seeds = torch.randperm(V, generator=gen)[:batch].to(torch.int32)
        out[f"s{step}_seeds"] = seeds.numpy()
        torch.manual_seed(seed + 10 + step)
        inp, outp, mfgs = sampler.sample_blocks(g, seeds)
        torch.manual_seed(seed + 10 + step)
        o_inp, _, o_blocks = bo.sample_blocks_bandit(og, seeds, fanouts, o_w, eta)
        assert torch.equal(o_inp, inp.long())
        rec = {}
        hooks = _hook_layers(net.gatv2_layers, rec)
        with torch.no_grad():
            x = mfgs[0].srcdata["features"]                         # train_lightning.py:138
            pred = net(mfgs, x)                                     # model.py:207-234 (stores embed_norm, a_ij on the blocks)
        for h in hooks:
            h.remove()
        o_pred, o_tr = bo.gatv2_forward(o_blocks, feats[o_inp], params, 0.2)
        out[f"s{step}_pred"] = bits(pred)
        assert np.array_equal(bits(o_pred), bits(pred)), f"{name}: oracle GATv2 output differs from the reference run"
        embed, aij = [], []
        for l, (b, ob) in enumerate(zip(mfgs, o_blocks)):
            check_block(ob, b, True, f"{name} step{step} layer{l}")
            block_arrays(f"s{step}_l{l}_", b, out, True)
            rst, e = rec[f"l{l}_out"]
            out[f"s{step}_l{l}_h_in"] = bits(rec[f"l{l}_in"])
            out[f"s{step}_l{l}_rst"] = bits(rst)                    # [S, H, D]
            out[f"s{step}_l{l}_e"] = bits(e.reshape(e.shape[0], -1))     # [B, H] pre-softmax logits, model.py:108-110
            out[f"s{step}_l{l}_a_ij"] = bits(b.edata["a_ij"])
            out[f"s{step}_l{l}_embed_norm"] = bits(b.srcdata["embed_norm"])
            for k_, v_ in (("rst", rst), ("e", e.reshape(e.shape[0], -1)), ("a_ij", b.edata["a_ij"]), ("embed_norm", b.srcdata["embed_norm"])):
                assert np.array_equal(bits(o_tr[l][k_]), bits(v_)), f"{name}: oracle {k_} differs at step {step} layer {l}"
            alpha = sampler.calculate_alpha(b)                      # bandit_sampler.py:146-154, the GAT branch
            out[f"s{step}_l{l}_alpha"] = bits(alpha)
            embed.append(b.srcdata["embed_norm"]); aij.append(b.edata["a_ij"])
        sampler.exp3(mfgs, g)                                       # train_lightning.py:469-471
        o_w, traces = bo.exp3(og, o_blocks, o_w, edge_w, embed, a_ij=aij)
        for l, (b, tr) in enumerate(zip(mfgs, traces)):
            assert np.array_equal(bits(tr["alpha"]), out[f"s{step}_l{l}_alpha"]), f"{name} alpha {step}/{l}"
            assert np.array_equal(bits(tr["rewards"]), bits(b.edata["rewards"])), f"{name} rewards {step}/{l}"
            out[f"s{step}_l{l}_rewards"] = bits(b.edata["rewards"])
        if not np.array_equal(bits(o_w), bits(sampler.exp3_weights)):
            raise SystemExit(f"{name}: exp3_weights differ from the reference after step {step}")
        out[f"s{step}_exp3_weights"] = bits(sampler.exp3_weights)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def sage_model_case(name, V, E, F, hidden, classes, fanouts, batch, eta, seed):
    """The reference's OWN model.SAGE.forward (model.py:312-333: embed_norm, layer call with edge_weight, activation) over the
    stand-in's SAGEConv ([DGL-recalled] semantics, oracle/dgl_standin.py) on blocks drawn by the reference sampler; bf16 on
    the CPU, relu, no dropout.  One step incl. the bandit update fed with the model's own embed_norm."""
    ip, ix, ei = chung_lu_csc(V, E, seed=seed)
    g = ref_graph(ip, ix, ei)
    og = bo.CSC(ip, ix, ei)
    feats = torch.randn(V, F, generator=torch.Generator().manual_seed(seed + 1)).bfloat16()
    g.ndata["features"] = feats
    edge_w = bo.normalized_edata(og)
    sampler = ref_bandit.PoissonBanditLadiesSampler(fanouts, importance_sampling=1, node_embedding="features", num_steps=1000, eta=eta, model="sage")
    torch.manual_seed(seed + 2)
    net = ref_model.SAGE(F, hidden, classes, len(fanouts), torch.relu, 0.0).bfloat16()
    out = dict(indptr=ip.numpy(), indices=ix.numpy(), eid=ei.numpy(), fanouts=np.array(fanouts), eta=np.array(eta),
               torch_seed=np.array(seed), features=bits(feats), hidden=np.array(hidden), classes=np.array(classes), edge_w=bits(edge_w))
    params = []
    for l, layer in enumerate(net.layers):
        out[f"p{l}_w_neigh"], out[f"p{l}_w_self"], out[f"p{l}_b_self"] = bits(layer.fc_neigh.weight.detach()), bits(layer.fc_self.weight.detach()), bits(layer.fc_self.bias.detach())
        params.append((layer.fc_self.weight.detach(), layer.fc_self.bias.detach(), layer.fc_neigh.weight.detach()))
    seeds = torch.randperm(V, generator=torch.Generator().manual_seed(seed + 3))[:batch].to(torch.int32)
    out["seeds"] = seeds.numpy()
    torch.manual_seed(seed + 10)
    inp, outp, mfgs = sampler.sample_blocks(g, seeds)
    torch.manual_seed(seed + 10)
    o_inp, _, o_blocks = bo.sample_blocks_bandit(og, seeds, fanouts, torch.ones(len(fanouts), og.num_edges, dtype=torch.bfloat16), eta)
    rec = {}
    hooks = _hook_layers(net.layers, rec)
    with torch.no_grad():
        pred = net(mfgs, mfgs[0].srcdata["features"])
    for h in hooks:
        h.remove()
    out["pred"] = bits(pred)
    embed = []
    for l, (b, ob) in enumerate(zip(mfgs, o_blocks)):
        check_block(ob, b, True, f"{name} layer{l}")
        block_arrays(f"l{l}_", b, out, True)
        out[f"l{l}_h_in"], out[f"l{l}_out"] = bits(rec[f"l{l}_in"]), bits(rec[f"l{l}_out"])
        out[f"l{l}_embed_norm"] = bits(b.srcdata["embed_norm"])
        embed.append(b.srcdata["embed_norm"])
        # the fp32 restatement (oracle.sage_conv_ref) stays within bf16 rounding of the reference's bf16 layer
        ws, bs, wn = params[l]
        ref32 = bo.sage_conv_ref(ob, rec[f"l{l}_in"], ws, bs, wn, ob.edge_weights)
        got = rec[f"l{l}_out"].float()
        assert (got - ref32).abs().max() <= 3 * ref32.abs().max() * 2 ** -8, f"{name}: layer {l} output vs fp32 restatement"
        en32 = bo.embed_norm_ref(rec[f"l{l}_in"])
        assert (b.srcdata["embed_norm"].float() - en32).abs().max() <= en32.abs().max() * 2 ** -8
    sampler.exp3(mfgs, g)
    o_w, traces = bo.exp3(og, o_blocks, torch.ones(len(fanouts), og.num_edges, dtype=torch.bfloat16), edge_w, embed)
    for l, (b, tr) in enumerate(zip(mfgs, traces)):
        assert np.array_equal(bits(tr["rewards"]), bits(b.edata["rewards"]))
        out[f"l{l}_rewards"] = bits(b.edata["rewards"])
    assert np.array_equal(bits(o_w), bits(sampler.exp3_weights))
    out["exp3_weights"] = bits(sampler.exp3_weights)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)



def row_checksum(w):
    """Order-sensitive checksum of a bf16 tensor's bit patterns (uint64 arithmetic, wraps)."""
    b = bits(w).astype(np.uint64).reshape(-1)
    idx = np.arange(1, b.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return np.uint64(((b + np.uint64(1)) * (idx * np.uint64(0x9E3779B97F4A7C15))).sum())


def collapse_case(name, V, E, F, fanouts, batch, eta, seed, max_steps=3000):
    """How long does the REFERENCE survive on N(0,1) features?  (VERDICT r2 item 1c / ADVICE r2: the Cora- and Pubmed-like
    bench windows stop after ~125 steps with a non-finite error -- the reference's arithmetic, or a product bug?)
    The reference's PoissonBanditLadiesSampler is run step after step -- sample_blocks, embed_norm = the real bf16 row norms
    of N(0,1) feature rows (width F) for the input layer and of relu(N(0,1)) rows (width 256) for the hidden layers, exp3 --
    until it raises.  With such norms every reward hits the cap (bandit_sampler.py:244), touched edges gain a factor e per
    step, F.normalize (:249) pushes everything else down by 1/e, and after a few hundred steps the untouched weights of a
    seed column leave bf16's range: the column sums to 0, w/0 = NaN (:131), and torch.bernoulli raises on the NaN
    probability (:423).  Fixture: the step at which the reference raised, a checksum of its EXP3 rows after every step,
    the rows themselves just before the end.  The oracle must follow bit for bit the whole way (subnormal sums included)."""
    ip, ix, ei = chung_lu_csc(V, E, seed=seed)
    g = ref_graph(ip, ix, ei)
    og = bo.CSC(ip, ix, ei)
    edge_w = bo.normalized_edata(og)
    feats = torch.randn(V, F, generator=torch.Generator().manual_seed(seed + 1)).bfloat16()
    norm0 = torch.norm(feats, dim=1)                                   # model.py:318-320 on the input rows
    hid = torch.relu(torch.randn(V, 256, generator=torch.Generator().manual_seed(seed + 2))).bfloat16()
    norm1 = torch.norm(hid, dim=1)
    sampler = ref_bandit.PoissonBanditLadiesSampler(fanouts, importance_sampling=1, node_embedding="features", num_steps=1000, eta=eta, model="sage")
    o_w = torch.ones(len(fanouts), og.num_edges, dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(seed + 3)
    sums, kept = [], []
    failed_at, how = -1, ""
    prev = None
    for step in range(max_steps):
        seeds = torch.randperm(V, generator=gen)[:batch].to(torch.int32)
        torch.manual_seed(seed + 1000 + step)
        try:
            inp, outp, mfgs = sampler.sample_blocks(g, seeds)
        except RuntimeError as ex:                                      # torch.bernoulli's range check on a NaN probability
            failed_at, how = step, str(ex).splitlines()[0][:120]
            torch.manual_seed(seed + 1000 + step)
            try:
                bo.sample_blocks_bandit(og, seeds, fanouts, o_w, eta)
                raise SystemExit(f"{name}: the reference raised at step {step} but the oracle did not")
            except (FloatingPointError, RuntimeError):
                pass
            break
        torch.manual_seed(seed + 1000 + step)
        o_inp, _, o_blocks = bo.sample_blocks_bandit(og, seeds, fanouts, o_w, eta)
        assert torch.equal(o_inp, inp.long()), f"{name}: input nodes differ at step {step}"
        embed = []
        for l, (b, ob) in enumerate(zip(mfgs, o_blocks)):
            check_block(ob, b, True, f"{name} step{step} layer{l}")
            en = (norm0 if l == 0 else norm1)[b.srcdata[NID].long()]
            b.srcdata["embed_norm"] = en
            embed.append(en)
        kept.append([b.num_src_nodes() for b in mfgs])
        prev = sampler.exp3_weights.clone()
        sampler.exp3(mfgs, g)
        o_w, _ = bo.exp3(og, o_blocks, o_w, edge_w, embed)
        if not np.array_equal(bits(o_w), bits(sampler.exp3_weights)):
            raise SystemExit(f"{name}: exp3_weights differ from the reference after step {step}")
        sums.append(row_checksum(sampler.exp3_weights))
    if failed_at < 0:
        raise SystemExit(f"{name}: the reference survived {max_steps} steps")
    w = sampler.exp3_weights.float()
    out = dict(indptr=ip.numpy(), indices=ix.numpy(), eid=ei.numpy(), fanouts=np.array(fanouts), eta=np.array(eta), batch=np.array(batch),
               torch_seed=np.array(seed), norm0=bits(norm0), norm1=bits(norm1), failed_at=np.array(failed_at), how=np.array(how),
               checksums=np.array(sums, dtype=np.uint64), kept=np.array(kept), last_weights=bits(sampler.exp3_weights),
               weights_before_last_update=bits(prev), min_positive=np.array([float(w[l][w[l] > 0].min()) for l in range(len(fanouts))]),
               zeros=np.array([int((w[l] == 0).sum()) for l in range(len(fanouts))]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "reference raised at step", failed_at, ":", how, "| smallest positive weights", out["min_positive"], "zeros", out["zeros"])


def toy():
    """ToyDataset, load_graph.py:96: edges 2->0, 3->0, 3->1, 4->1, + self loops (train_lightning.py:334-335)."""
    indptr = torch.tensor([0, 3, 6, 7, 8, 9])
    indices = torch.tensor([2, 3, 0, 3, 4, 1, 2, 3, 4], dtype=torch.int32)
    eid = torch.tensor([0, 1, 4, 2, 3, 5, 6, 7, 8], dtype=torch.int32)
    return indptr, indices, eid


def main():
    ip, ix, ei = toy()
    bandit_case("toy_poisson_bandit", ip, ix, ei, [torch.tensor([0, 1], dtype=torch.int32)] * 2, [2], 0.1, 0)
    bandit_case("toy_poisson_bandit_3layer", ip, ix, ei, [torch.tensor([0, 1], dtype=torch.int32)] * 2, [3, 2, 2], 0.4, 3)
    for i, (V, E, fan, batch, eta) in enumerate([(300, 3000, [40, 20, 10], 8, 0.1),
                                                 (1000, 20000, [128, 64, 32], 16, 0.1),
                                                 (2000, 12000, [512, 256, 128], 32, 0.4)]):
        ip, ix, ei = chung_lu_csc(V, E, seed=10 + i)
        gen = torch.Generator().manual_seed(2 + i)
        steps = [torch.randperm(V, generator=gen)[:batch].to(torch.int32) for _ in range(3)]
        bandit_case(f"synth{i}_poisson_bandit", ip, ix, ei, steps, fan, eta, 100 + i)
        ladies_case(f"synth{i}_poisson_ladies", ip, ix, ei, steps[0], fan, 200 + i)
    ip, ix, ei = chung_lu_csc(300, 3000, seed=10)
    gen = torch.Generator().manual_seed(9)
    steps = [torch.randperm(300, generator=gen)[:8].to(torch.int32) for _ in range(2)]
    bandit_case("synth0_bandit_multinomial", ip, ix, ei, steps, [40, 20, 10], 0.1, 300, poisson=False)
    bandit_case("synth0_poisson_bandit_uniform_nodes", ip, ix, ei, steps, [40, 20, 10], 0.1, 302, importance_sampling=0)
    ladies_case("synth0_ladies_multinomial", ip, ix, ei, steps[0], [40, 20, 10], 301, poisson=False)
    # LadiesSampler(importance_sampling=False), ladies_sampler.py:49-51: fp32 ones as importances (unreachable from the CLI,
    # train_lightning.py:360 passes only the fanouts; CPU tensors only -- torch.ones(...) there has no device)
    ladies_case("synth0_ladies_multinomial_uniform_nodes", ip, ix, ei, steps[0], [40, 20, 10], 303, poisson=False, importance_sampling=False)
    # round 3: the model side run from the reference's own model.py
    gat_case("gat0_model_exp3", 600, 7000, 24, 16, 5, [4, 4, 1], [80, 40, 20], 12, 0.1, True, 2, 400)
    gat_case("gat1_model_exp3_noresidual", 400, 5000, 32, 8, 3, [2, 2, 1], [60, 30, 15], 8, 0.4, False, 2, 410)
    sage_model_case("sage0_model_exp3", 600, 7000, 40, 16, 5, [80, 40, 20], 12, 0.1, 420)
    collapse_case("collapse0_normal_features", 400, 1600, 1433, [64, 32, 16], 8, 0.1, 50)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "collapse":
        collapse_case("collapse0_normal_features", 400, 1600, 1433, [64, 32, 16], 8, 0.1, 50)
    else:
        main()
