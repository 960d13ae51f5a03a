import sys, types, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import os
from conftest import load_gold, rel_l2
for prec in ('f32','f16x3'):
    os.environ['SCIPNP_CONV_PRECISION']=prec
    from adaptivepnp_sci_amd import ddnet_train
    from adaptivepnp_sci_amd import test_ddnet as plug
    from oracle.nets import cpu_data_parallel, synth_ddnet_weights
    from oracle.sci_ops import one_to_three_channel
    for gold in ('ddnet_finetune_32x48x8','ddnet_finetune_64x64x8'):
        g=load_gold(gold); net=cpu_data_parallel(synth_ddnet_weights(0)); grads={}
        ddnet_train.GRAD_HOOK=grads.update
        args=types.SimpleNamespace(dm_update=True, dm_lr=float(g['lr']), dm_update_per_iter=1)
        plug(one_to_three_channel(torch.from_numpy(g['mosaic'])).cuda(), None, None, net, True, args)
        ddnet_train.GRAD_HOOK=None
        rows=[]
        for k,gv in grads.items():
            key=k.replace('.','_')
            nerr=abs(float(torch.linalg.vector_norm(gv.double()))/float(g['gradnorm_'+key])-1)
            e=rel_l2(gv.cpu().numpy(), g['grad_'+key]) if 'grad_'+key in g.files else float('nan')
            rows.append((e,nerr,k))
        rows.sort(key=lambda r:-(r[0] if r[0]==r[0] else -1))
        print(prec,gold,'worst full-tensor errs:',[(f'{e:.1e}',k) for e,_,k in rows[:5]], 'worst norm err', max(r[1] for r in rows))
