import sys, torch, numpy as np, torch.nn.functional as F
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_kernels_gpu as T
from multitalent_amd import ops
dev=torch.device('cuda',0)
ops.set_option('conv_wino',2)
N,Cin,Cout,shape,two_src=1,30,30,(5,7,19),True
g=torch.Generator().manual_seed(21)
srcs=[torch.randn((N,Cin)+shape,generator=g)]
lazy=[(torch.rand((N,Cin),generator=g)+0.5, torch.randn((N,Cin),generator=g),0.01)]
srcs.append(torch.randn((N,Cin)+shape,generator=g)); lazy.append(None)
Ct=60
w=torch.randn((Cout,Ct,3,3,3),generator=g)/np.sqrt(Ct*27)
b=torch.randn(Cout,generator=g)
xin=T.ref_inputs(srcs,lazy)
x=xin.clone().requires_grad_(True)
y=F.conv3d(x,w,None,padding=1)
dy=torch.randn(y.shape,generator=g)
y.backward(dy)
print('dy shape',dy.shape,'x.grad',x.grad.shape)
C0=30
base0=torch.randn((N,)+shape+(C0,),generator=g); base1=torch.randn((N,)+shape+(Ct-C0,),generator=g)
d0,d1=base0.to(dev),base1.to(dev)
geomT=ops.ConvGeom(shape,(3,3,3),(1,1,1),(1,1,1))
dyd=T.to_ndhwc(dy).to(dev)
p=ops.fill_conv([ops.Act(dyd)],geomT,Ct,out0=ops.Act(d0),out1=ops.Act(d1),csplit=C0,accumulate=True)
print(ops.conv_kernel_name(p), 'Cin',p.Cin,'Cout',p.Cout,'csplit',p.csplit)
wd=w.to(dev).contiguous()
wp=ops.pack_conv_weights(wd,Cout,0,Ct,(3,3,3),ops.conv_weight_strides(wd,as_bwd_data=True),True,ops.conv_ck(p),layout=ops.conv_pack_layout(p))
p.wpack=wp.data_ptr()
ops.conv3d_fwd(p); torch.cuda.synchronize()
got=torch.cat([d0.cpu()-base0, d1.cpu()-base1],-1)
e=(T.to_ncdhw(got)-x.grad).abs()
print('relerr',float(e.max()/x.grad.abs().max()),[round(float(e[:,c].max()),3) for c in range(0,60,6)])
