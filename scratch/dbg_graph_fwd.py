import sys; sys.path.insert(0,'.')
import torch, gc
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.nn import embed_norm, weighted_aggregate
from bliss_gnn_amd.synth import chung_lu_csc
from bliss_gnn_amd.train import BatchLoader, GraphedTrainStep
def log(*a): print(*a, flush=True)
level=sys.argv[1]
cuda=torch.device('cuda:0')
ip, ix, ei = chung_lu_csc(8000, 160000, seed=12)
feats = torch.randn(8000, 64, generator=torch.Generator().manual_seed(2)).bfloat16()
labels = torch.randint(0, 5, (8000,), generator=torch.Generator().manual_seed(3))
fan, bs = [400, 200, 100], 64
ids = torch.arange(8000, dtype=torch.int32, device=cuda)
g = bg.Graph(ip.to(cuda), ix.to(cuda), ei.to(cuda), ndata={"features": feats.to(cuda), "labels": labels.to(cuda)})
g.edata["w"] = bg.normalized_edata(g)
sampler = bg.PoissonBanditLadiesSampler(fan, eta=0.1)
torch.manual_seed(0)
model = SAGE(64, 32, 5, 3, torch.relu, 0.0).to(cuda).bfloat16()
gs = GraphedTrainStep(g, sampler, model, bs)
l2 = BatchLoader(ids, bs, seed=5).forever()
torch.manual_seed(9)
gs.calibrate(l2, steps=3)
import contextlib
keep=[]
nl = 3 if level.startswith('V') else (2 if level=='C2n' else int(level[1])); nograd = level.endswith('n')
def body():
    ctx = torch.no_grad() if nograd else contextlib.nullcontext()
    with ctx:
        inp,outp,mfgs = sampler.sample_blocks_static(g, gs.seeds)
        h = mfgs[0].srcdata["features"]
        if level == 'C2n':
            b2 = mfgs[2]
            snapA = [b2.indptr.clone(), b2.src.clone(), b2.dst.clone(), b2._counts_dev.clone(), b2.edata["edge_weights"].clone()]
        for l in range(min(nl,3)):
            blk = mfgs[l]
            blk.srcdata["embed_norm"] = embed_norm(h)
            if l == 2 and level.startswith('V'):
                lay = model.layers[2]
                if level == 'V1n':      # GEMMs only, no SpMM
                    h = lay.fc_self(h[:blk.num_dst_nodes()]) + lay.fc_neigh(h)[:blk.num_dst_nodes()]
                elif level == 'V2n':    # scalar-path SpMM only, no GEMM
                    h = weighted_aggregate(blk, h[:, :5].contiguous(), blk.edata["edge_weights"], mean=True)
                elif level == 'V3n':    # GEMM feeding the SpMM, no fc_self
                    h = weighted_aggregate(blk, lay.fc_neigh(h), blk.edata["edge_weights"], mean=True)
            else:
                h = model.layers[l](blk, h, edge_weight=blk.edata["edge_weights"])
            if l < 2: h = torch.relu(h)
        if level == 'C2n':
            b2 = mfgs[2]
            snapB = [b2.indptr.clone(), b2.src.clone(), b2.dst.clone(), b2._counts_dev.clone(), b2.edata["edge_weights"].clone()]
            keep[:] = [snapA, snapB]
        if nl == 4:
            y = mfgs[-1].dstdata["labels"]
            return gs.loss_fn(h, y).detach()
        return h.float().sum().detach()
gs._body = body
gs.capture(l2, warmup=3)
log('level',level,'captured', [(c.S,c.E,c.C,c.K,c.B,c.err) for c in gs.last_counts])
for i in range(6):
    gs(next(l2)); log('level',level,'replay', i, float(gs.loss), [(c.K,c.B,c.err) for c in gs.last_counts])
    if level == 'C2n':
        A,B_ = keep
        c0 = gs.last_counts[0]
        log('  block2 counts dev', A[3].tolist(), 'B', c0.B, 'S', c0.S)
        names=['indptr','src','dst','counts','w']
        for nm,a,b in zip(names,A,B_):
            log('   ', nm, 'changed between start and end of fwd:', int((a!=b).sum()))
        log('   indptr', A[0][:8].tolist(), '...', A[0][-3:].tolist(), 'src max', int(A[1][:c0.B].max()), 'dst max', int(A[2][:c0.B].max()), 'src tail', A[1][c0.B:c0.B+4].tolist())
log('level',level,'done')
