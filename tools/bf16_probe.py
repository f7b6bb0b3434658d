import sys, numpy as np, torch
sys.path[:0]=['.','tests/golden','tests']
from synth import GRAFP_CFG, synth_state
from neuralsampleid_amd import ops, functional as F_
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.simclr.simclr import SimCLR
from neuralsampleid_amd.simclr.ntxent import ntxent_loss
import conftest
g=conftest.load_golden('e2e_b8_k3')
def rel(a,b):
    a=a.detach().cpu().double(); b=torch.from_numpy(b).double(); return float((a-b).norm()/b.norm())
def cosmin(a,b):
    a=a.detach().cpu().double(); b=torch.from_numpy(b).double(); return float(torch.nn.functional.cosine_similarity(a,b,dim=1).min())
for prec in ('fp32','bf16'):
    ops.set_gemm_precision(prec)
    for mode,tag in (('eval','eval'),('train','s0')):
        model=SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG,in_channels=8,k=3,size='t'))
        model.load_state_dict(synth_state(model.state_dict())); model.cuda()
        model.train(mode=='train')
        n=len([k for k in g.files if k.startswith(f'knn.{tag}.')])
        for forced in (True, False):
            F_.TAPE=F_.KnnTape(replay=[torch.from_numpy(g[f'knn.{tag}.{c}']) for c in range(n)] if forced else None)
            with torch.no_grad():
                h_i,h_j,z_i,z_j=model(torch.from_numpy(g['x_i']).cuda(), torch.from_numpy(g['x_j']).cuda())
                loss=float(ntxent_loss(z_i,z_j,GRAFP_CFG))
            own=F_.TAPE.recorded; F_.TAPE=None
            sfx='eval' if mode=='eval' else 'train'
            agree=np.mean([(np.sort(a.cpu().numpy(),-1)==np.sort(g[f'knn.{tag}.{c}'],-1)).all(-1).mean() for c,a in enumerate(own)])
            lref=float(g['loss_eval'][0]) if mode=='eval' else float(g['losses'][0])
            print(f"{prec} {mode:5s} forced={forced!s:5s} loss {loss:.4f} (ref {lref:.4f}) rel_h {rel(h_i,g['h_i_'+sfx]):.2e} rel_z {rel(z_i,g['z_i_'+sfx]):.2e} min cos z {cosmin(z_i,g['z_i_'+sfx]):.5f} knn set agreement {agree:.4f}")
