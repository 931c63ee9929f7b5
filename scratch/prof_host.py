import sys, time, cProfile, pstats; sys.path.insert(0,'.')
import torch
import bliss_gnn_amd as bg
from bliss_gnn_amd.model import SAGE
from bliss_gnn_amd.synth import CONFIGS, chung_lu_csc, node_data
from bliss_gnn_amd.train import BatchLoader, TrainStep
dev=torch.device('cuda:0'); cfg=CONFIGS['reddit']
ip,ix,ei=chung_lu_csc(cfg['num_nodes'],cfg['num_edges'],seed=0,device=dev)
feats,labels,train_nid=node_data(cfg['num_nodes'],cfg['feat'],cfg['classes'],cfg['n_train'],seed=1,device=dev)
g=bg.Graph(ip,ix,ei,ndata={'features':feats,'labels':labels}); g.edata['w']=bg.normalized_edata(g)
s=bg.PoissonBanditLadiesSampler(cfg['fanouts'],eta=0.1)
model=SAGE(cfg['feat'],256,cfg['classes'],3,torch.relu,0.1).to(dev).bfloat16()
step=TrainStep(g,s,model); loader=BatchLoader(train_nid,256).forever()
for _ in range(5): step(next(loader))
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
t=time.perf_counter()
for _ in range(10): step(next(loader))
torch.cuda.synchronize(); dt=time.perf_counter()-t
pr.disable()
print('ms/step',dt*100)
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
