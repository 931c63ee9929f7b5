import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, ctypes as C
from conftest import bf16_bits, bits_to_bf16, load_golden
import bliss_gnn_amd as bg
from bliss_gnn_amd import _lib
from oracle import bliss_oracle as bo, numerics as nx
z=load_golden('synth0_poisson_bandit'); dev=torch.device('cuda:0')
ip, ix, ei = torch.from_numpy(z["indptr"]), torch.from_numpy(z["indices"]), torch.from_numpy(z["eid"])
g = bg.Graph(ip.to(dev), ix.to(dev), ei.to(dev)); g.edata['w']=bg.normalized_edata(g)
og=bo.CSC(ip,ix,ei); edge_w=bo.normalized_edata(og)
fan=z['fanouts'].tolist(); eta=float(z['eta']); seed=int(z['torch_seed'])
s=bg.PoissonBanditLadiesSampler(fan,eta=eta)
o_w=torch.ones(3,og.num_edges,dtype=torch.bfloat16)
for step in range(3):
    seeds=torch.from_numpy(z[f's{step}_seeds'])
    torch.manual_seed(seed+step); inp,_,blocks=s.sample_blocks(g,seeds.to(dev))
    torch.manual_seed(seed+step); _,_,ob=bo.sample_blocks_bandit(og,seeds,fan,o_w,eta)
    emb=[]
    for l,b in enumerate(blocks):
        en=bits_to_bf16(z[f's{step}_l{l}_embed_norm']); b.srcdata['embed_norm']=en.to(dev); emb.append(en)
    before=s.exp3_weights.clone()
    s.exp3(blocks,g); torch.cuda.synchronize()
    o_w2,tr=bo.exp3(og,ob,o_w,edge_w,emb)
    for l in range(3):
        mine=s.exp3_weights[l].cpu(); ref=o_w2[l]
        nd=(mine.view(torch.int16)!=ref.view(torch.int16)).sum().item()
        print('step',step,'layer',l,'ndiff',nd,'norm gpu',s._norms[l].item(),'norm ref',tr[l]['norm'].item(), 'rowsum', s._row_sum[l].tolist(), 'scratch', s._scratch[l].tolist())
        if nd:
            idx=(mine.view(torch.int16)!=ref.view(torch.int16)).nonzero()[:5,0]
            print(' idx',idx.tolist(),'mine',mine[idx].float().tolist(),'ref',ref[idx].float().tolist(),'before',before[l].cpu()[idx].float().tolist())
            # unnormalised reference
            wtmp=o_w[l].clone(); wtmp[ob[l].eid]=wtmp[ob[l].eid]*tr[l]['exp_rewards']
            print(' ref unnormalised at idx', wtmp[idx].float().tolist(), 'exact sum ref', nx.row_exact_sum(wtmp)/2**64)
    o_w=o_w2
