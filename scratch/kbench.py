import sys, time; sys.path.insert(0,'.')
import torch
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader, TrainStep
from bliss_gnn_amd import roofline
dev=torch.device('cuda:0'); cfg=CONFIGS[sys.argv[1] if len(sys.argv)>1 else 'reddit']
ip,ix,ei=chung_lu_csc(cfg['num_nodes'],cfg['num_edges'],seed=0,device=dev)
feats,labels,train_nid=node_data(cfg['num_nodes'],cfg['feat'],cfg['classes'],cfg['n_train'],seed=1,device=dev)
g=bg.Graph(ip,ix,ei,ndata={'features':feats,'labels':labels}); g.edata['w']=bg.normalized_edata(g)
s=bg.PoissonBanditLadiesSampler(cfg['fanouts'],eta=0.1)
model=SAGE(cfg['feat'],256,cfg['classes'],3,torch.relu,0.1).to(dev).bfloat16()
step=TrainStep(g,s,model); loader=BatchLoader(train_nid,cfg['batch']).forever()
for _ in range(10): step(next(loader))
torch.cuda.synchronize()
N=30
# whole step
t=time.perf_counter()
for _ in range(N): step(next(loader))
torch.cuda.synchronize(); print('step ms', (time.perf_counter()-t)/N*1e3)
# sampling only
t=time.perf_counter()
for _ in range(N): s.sample(g,next(loader))
torch.cuda.synchronize(); print('sample ms', (time.perf_counter()-t)/N*1e3)
# model fwd/bwd only on fixed blocks
inp,outp,blocks=s.sample(g,next(loader))
x=blocks[0].srcdata['features']; y=blocks[-1].dstdata['labels']
lossf=torch.nn.CrossEntropyLoss()
def mstep():
    pred=model(blocks,x); loss=lossf(pred,y); step.opt.zero_grad(set_to_none=True); loss.backward(); step.opt.step()
for _ in range(5): mstep()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(N): mstep()
torch.cuda.synchronize(); print('model fwd/bwd/adam ms', (time.perf_counter()-t)/N*1e3)
t=time.perf_counter()
for _ in range(N): s.exp3(blocks,g)
torch.cuda.synchronize(); print('exp3 ms', (time.perf_counter()-t)/N*1e3)
tm=roofline.KernelTimer(); tm.enable('all')
for _ in range(N): step(next(loader))
torch.cuda.synchronize()
r=tm.read(); tm.enable('off')
tot=sum(v['total_ms'] for v in r.values())
print('library kernels ms/step', tot/N)
for k,v in sorted(r.items(), key=lambda kv:-kv[1]['total_ms']):
    print(f"{k:22s} launches/step={v['launches']/N:5.1f} avg_us={v['avg_us']:8.2f} ms/step={v['total_ms']/N:7.3f}")
