"""CPU: how well is the FastDVDnet finetune gradient determined in fp32?  The oracle network (bit-exact restatement of the
reference, synthetic weights) evaluated in float32 and in float64 on the same inputs: the fp32 gradient of most layers
deviates from the fp64 one by 0.4-1.5e-4 relative L2 (ReLU masks flip where an activation is within round-off of 0), the
layers behind the last ReLU-free path by 2e-7.  This bounds the agreement ANY two fp32 implementations can show
(tests/test_gpu_solver.py::test_fastdvdnet_online_finetune_matches_reference uses 1e-3 for that reason)."""
import sys, copy
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import nets as ON, denoisers as OD, sci_ops as OO
from adaptivepnp_sci_amd import synth
torch.set_num_threads(8)
y, Phi, orig = synth.make_problem(64,64,8,seed=5)
rng=np.random.default_rng(1)
v = torch.from_numpy(np.clip(np.repeat(orig[:,:,None,:],3,2)+0.05*rng.standard_normal((64,64,3,8)),0,1).astype(np.float32))
yp = OO.bayer_split(torch.from_numpy(y)); Pp = OO.bayer_split(torch.from_numpy(Phi))
noise = rng.normal(0,5/255,(8,3,64,64))
def grads(dtype):
    net = torch.nn.DataParallel(ON.synth_fastdvdnet_weights(0)) if False else ON.cpu_data_parallel(ON.synth_fastdvdnet_weights(0))
    net = net.to(dtype)
    # replicate fastdvdnet_pass first step in dtype
    import torch.nn as nn
    vv = v.permute(3,2,0,1).to(dtype)
    v_plus = vv + torch.from_numpy(vv.numpy().astype(np.float64)+noise).to(torch.float32).to(dtype)
    Phi_m = OO.bayer_merge(Pp).to(dtype); y_m = OO.bayer_merge(yp).to(dtype)
    net.train()
    for m in net.module.modules():
        if isinstance(m, nn.BatchNorm2d): m.eval()
    N,C,H,W = vv.shape
    nm = torch.tensor([8/255],dtype=dtype).expand((1,1,H,W))
    den = torch.empty((N,C,H,W),dtype=dtype)
    for n in range(N):
        idx=(torch.arange(n,n+5)-2)%N
        den[n]=net(v_plus[idx].reshape((1,-1,H,W)), nm)
    den=den.permute(2,3,1,0)
    loss = nn.MSELoss()(torch.sum(OD._rgb_cube_to_mosaic(den).to(dtype)*Phi_m,dim=2), y_m)
    loss.backward()
    return {n:p.grad.double().clone() for n,p in net.named_parameters()}, float(loss)
g32,l32 = grads(torch.float32)
g64,l64 = grads(torch.float64)
print('loss',l32,l64)
for k in g32:
    if k.endswith('weight') and g32[k].dim()==4:
        e=float((g32[k]-g64[k]).norm()/g64[k].norm())
        print(f'{k:55s} fp32-vs-fp64 rel {e:.2e}')
