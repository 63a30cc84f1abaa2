python - <<'PY'
import torch, numpy as np, torch.nn.functional as F
import sys; sys.path.insert(0,'.')
from multitalent_amd import ops
dev=torch.device('cuda',0)
def run(N,Cin,Cout,shape,lazy=False,stats=True):
    g=torch.Generator().manual_seed(1)
    x=torch.randn((N,Cin)+shape,generator=g); w=torch.randn((Cout,Cin,3,3,3),generator=g)/np.sqrt(Cin*27); b=torch.randn(Cout,generator=g)
    xb=x.permute(0,2,3,4,1).contiguous().to(dev)
    act=ops.Act(xb)
    xin=x
    if lazy:
        sc=(torch.rand((N,Cin),generator=g)+0.5); sh=torch.randn((N,Cin),generator=g)
        act=ops.Act(xb,scale=sc.to(dev).contiguous(),shift=sh.to(dev).contiguous(),slope=0.01)
        xin=F.leaky_relu(x*sc[:,:,None,None,None]+sh[:,:,None,None,None],0.01)
    geom=ops.ConvGeom(shape,(3,3,3),(1,1,1),(1,1,1))
    out=torch.full((N,)+geom.out+(Cout,),float('nan'),device=dev)
    bd=b.to(dev)
    p=ops.fill_conv([act],geom,Cout,out0=ops.Act(out),bias=bd)
    name=ops.conv_kernel_name(p)
    wd=w.to(dev).contiguous()
    wp=ops.pack_conv_weights(wd,Cin,0,Cout,(3,3,3),ops.conv_weight_strides(wd),False,ops.conv_ck(p),layout=ops.conv_pack_layout(p))
    p.wpack=wp.data_ptr()
    part=torch.zeros((N,ops.conv_stats_blocks(p),Cout,2),device=dev); p.stats_part=part.data_ptr()
    ops.conv3d_fwd(p); torch.cuda.synchronize()
    ref=F.conv3d(xin,w,b,padding=1)
    got=out.permute(0,4,1,2,3).cpu()
    err=float((got-ref).abs().max()/ref.abs().max())
    s=part.cpu().double().sum(1)
    serr=float((s[...,0]-ref.double().sum((2,3,4))).abs().max())
    print(name,(N,Cin,Cout,shape),'relerr %.2e'%err,'stat err %.2e'%serr, 'nan' if torch.isnan(got).any() else '')
run(2,32,32,(8,16,64))
run(2,30,30,(9,18,70),lazy=True)
run(1,64,40,(12,20,33),lazy=True)
run(2,16,64,(24,48,64))
PY
python tools/bench_conv.py --mode fwd --cin 32 --cout 32 --reps 5 | tail -1
python tools/bench_conv.py --mode fwd --cin 64 --cout 32 --reps 5 | tail -1
python tools/bench_conv.py --mode fwd --cin 64 --cout 64 --shape 24 96 96 --reps 5 | tail -1
