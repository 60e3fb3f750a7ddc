import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/iccv2025-gdl_amd'); sys.path.insert(0,'/root/repo/tests')
from oracle import fixtures as fx
from gdl.swin import SwinEngine
cfg=fx.SWIN_TINY2; DEV='cuda:0'
for dtype in ('f32','bf16'):
    eng=SwinEngine(cfg,dtype,2,2,DEV)
    P=fx.make_state(fx.swin_param_shapes(cfg))
    params=[torch.from_numpy(v).to(DEV) for v in P.values()]
    eng.set_params(params)
    x=torch.from_numpy(fx.swin_input(cfg,2,2,0)).to(DEV)
    y=eng.forward(x); torch.cuda.synchronize()
    print(dtype,'y finite',torch.isfinite(y).all().item(), float(y.abs().mean()))
    def chk(name,t): print('   ',name, torch.isfinite(t.float()).all().item(), float(t.float().abs().mean()))
    chk('pe_rows',eng.pe_rows); chk('pe_out',eng.pe_out); chk('x0',eng.x0)
    b=eng.stages[0]['blocks'][0]
    for k in ('h','qkv_a','attn','x_mid','m','u','a','x_out'): chk(k,b[k])
    grads=[torch.full_like(p,float('nan')) for p in params]
    eng.backward(torch.randn(4,192,device=DEV),grads); torch.cuda.synchronize()
    bad=[n for (n,_),g in zip(eng.param_shapes(),grads) if not torch.isfinite(g).all()]
    print('  bad grads:',len(bad), bad[:8])
