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


if __name__ == "__main__":
    main()
