"""Two real ranks on ONE GPU (gloo carries the collectives through the host): the static-shape replica step -- gradient
all-reduce, packed EXP3 all-gather, bliss_exp3_apply_ranks over both ranks' lists -- must leave bit-identical EXP3 rows
and parameters on both ranks although every rank samples and trains on its own batches."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import bliss_gnn_amd as bg
    from bliss_gnn_amd import dist as bdist
    from bliss_gnn_amd.model import SAGE
    from bliss_gnn_amd.synth import chung_lu_csc
    from bliss_gnn_amd.train import BatchLoader, GraphedTrainStep
    ip, ix, ei = chung_lu_csc(5000, 90000, seed=21)
    feats = torch.randn(5000, 32, generator=torch.Generator().manual_seed(2)).bfloat16()
    labels = torch.randint(0, 4, (5000,), generator=torch.Generator().manual_seed(3))
    g = bg.Graph(ip.to(dev), ix.to(dev), ei.to(dev), ndata={"features": feats.to(dev), "labels": labels.to(dev)})
    g.edata["w"] = bg.normalized_edata(g)
    s = bg.PoissonBanditLadiesSampler([300, 150, 80], eta=0.1)
    torch.manual_seed(0)
    model = SAGE(32, 16, 4, 3, torch.relu, 0.0).to(dev).bfloat16()
    bdist.broadcast_parameters(model)
    ids = torch.arange(5000, dtype=torch.int32, device=dev)
    loader = BatchLoader(ids, 48, seed=5 + rank).forever()          # every rank its own batches ...
    torch.manual_seed(9 + rank)                                      # ... and its own sampler stream
    step = GraphedTrainStep(g, s, model, 48, distributed=True)
    step.calibrate(loader, steps=3)
    for _ in range(4):                                               # static shapes, launched kernel by kernel (gloo cannot be captured)
        loss = step.eager_step(next(loader))
        s.check_errors()
    torch.cuda.synchronize()
    torch.save(dict(w=s.exp3_weights.cpu().view(torch.int16), params=[p.detach().cpu() for p in model.parameters()],
                    loss=float(loss)), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_keep_identical_bandit_state(cuda):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker, args=(r, world, port, outdir)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
        a, b = (torch.load(os.path.join(outdir, f"r{r}.pt")) for r in range(world))
    assert torch.equal(a["w"], b["w"])                               # the replicas' EXP3 rows: bit-identical
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.equal(pa, pb)                                   # and so are the parameters (same averaged gradients)
    assert a["loss"] != b["loss"]                                    # (they did train on different batches)
    ones = torch.ones_like(a["w"].view(torch.bfloat16)).view(torch.int16)
    assert not torch.equal(a["w"], ones)
