import sys, torch
sys.path[:0]=['.','tests/golden','tests']
from synth import GRAFP_CFG
from neuralsampleid_amd import ops
from neuralsampleid_amd.encoder.graph_encoder import GraphEncoder
from neuralsampleid_amd.simclr.simclr import SimCLR
from neuralsampleid_amd.simclr.ntxent import ntxent_loss
from neuralsampleid_amd.optim import FusedClipAdam
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
gi=torch.Generator().manual_seed(0); gj=torch.Generator().manual_seed(1)
x_i=(torch.randn(B,64,128,generator=gi)*20-40); x_j=(x_i+3*torch.randn(B,64,128,generator=gj))
x_i,x_j=x_i.cuda(),x_j.cuda()
curves={}
for prec in ('fp32','bf16','fp32b'):
    ops.set_gemm_precision('bf16' if prec=='bf16' else 'fp32')
    torch.manual_seed(42 if prec!='fp32b' else 43)
    model=SimCLR(GRAFP_CFG, GraphEncoder(GRAFP_CFG,in_channels=8,k=3,size='t')).cuda().train()
    opt=FusedClipAdam(model.parameters(), lr=8e-5, max_norm=1.0)
    ls=[]
    for step in range(40):
        opt.zero_grad(); _,_,z_i,z_j=model(x_i,x_j); loss=ntxent_loss(z_i,z_j,GRAFP_CFG); loss.backward(); opt.step(); ls.append(float(loss))
    curves[prec]=ls
    print(prec, ' '.join(f'{v:.3f}' for v in ls[::3]))
import math
print('max |log ratio| bf16 vs fp32:', max(abs(math.log(a/b)) for a,b in zip(curves['bf16'],curves['fp32'])))
print('max |log ratio| fp32 seed43 vs fp32 seed42:', max(abs(math.log(a/b)) for a,b in zip(curves['fp32b'],curves['fp32'])))
