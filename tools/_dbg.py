import sys; sys.path.insert(0,'.')
import torch
from fusion_gcn_amd import ops
torch.manual_seed(0)
for (B,T,V,K,N) in ((1,4,25,32,64),(1,4,25,32,128),(2,20,25,64,64)):
    a=torch.randn(B,T,V,K,device='cuda'); g=torch.randn(B,T,V,N,device='cuda')
    got=ops.tconv_wgrad(a,g,taps=9,stride=1)
    ref=ops.rows_wgrad(a,g,K=K,N=N,tmap=ops.conv_tmap(9,1))
    print(B,T,V,K,N,'err',((got-ref).norm()/ref.norm()).item())
    print(' got[4,:3,:4]',got[4,:3,:4].cpu().numpy().round(3).tolist())
    print(' ref[4,:3,:4]',ref[4,:3,:4].cpu().numpy().round(3).tolist())
    # which entries match?
    m=(got-ref).abs()<1e-3*ref.abs().max()
    print(' match frac per tap',m.float().mean(dim=(1,2)).cpu().numpy().round(2).tolist())
    print(' match frac per k (tap4)',m[4].float().mean(dim=1).cpu().numpy().round(2).tolist()[:40])
