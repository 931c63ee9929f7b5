"""CPU suite: the N>1 exchange logic of bliss_gnn_amd/dist.py on world_size-2 gloo (no GPU needed)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bliss_gnn_amd import dist as bdist
    torch.manual_seed(rank)
    # 1) flat-bucket gradient averaging
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    bdist.broadcast_parameters(model)
    w0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    for p in model.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    bdist.allreduce_gradients(model)
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # 2) ragged update exchange, rank order, identical everywhere
    n = 3 + 4 * rank
    pos = (torch.arange(n, dtype=torch.int32) * 10 + rank)
    fac = (torch.arange(n, dtype=torch.float32) * 0.01 + 1 + rank).bfloat16()
    got = bdist.gather_updates(pos, fac)
    torch.save((rank, w0, g, [(a.clone(), b.float().clone()) for a, b in got]), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradient_bucket_and_update_gather():
    world, port = 2, _free_port()
    import tempfile
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker, args=(r, world, port, outdir)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
        res = [torch.load(os.path.join(outdir, f"r{r}.pt")) for r in range(world)]
    (_, w_a, g_a, u_a), (_, w_b, g_b, u_b) = res
    assert torch.equal(w_a, w_b)                                   # broadcast made the replicas identical
    assert torch.allclose(g_a, torch.full_like(g_a, 1.5)) and torch.equal(g_a, g_b)   # mean of 1 and 2
    assert len(u_a) == 2
    for (pa, fa), (pb, fb) in zip(u_a, u_b):
        assert torch.equal(pa, pb) and torch.equal(fa, fb)          # identical on every rank
    for r, (p, f) in enumerate(u_a):                                # rank order, exact ragged lengths
        n = 3 + 4 * r
        assert p.tolist() == [i * 10 + r for i in range(n)]
        assert torch.equal(f, (torch.arange(n, dtype=torch.float32) * 0.01 + 1 + r).bfloat16().float())
