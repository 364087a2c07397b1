import torch, sys, math
sys.path.insert(0,'.')
import ideas_amd.op as op
CL=torch.channels_last
torch.manual_seed(27)
B,ci,co,H=2,3,16,10
x=torch.randn(B,ci,H,H).cuda().contiguous(memory_format=CL)
w=torch.randn(co,ci,1,1).cuda().contiguous(memory_format=CL)
b=(torch.randn(co)*0.3).cuda()
gain=1/math.sqrt(ci)
yf=op.conv2d_bias_act(x,w,b,gain=gain)
yc=op.conv2d(x,w,None,gain=gain)
yu=op.fused_leaky_relu(yc,b)
d=(yf-yu).abs()
print('max diff',float(d.max()),'n diff',int((d>0).sum()),'of',d.numel())
i=int(d.flatten().argmax())
print(float(yf.flatten()[i]),float(yu.flatten()[i]),float(yc.flatten()[i]))
# cpu emulate
yc_c=yc.cpu(); b_c=b.cpu()
v=yc_c+b_c.view(1,-1,1,1)
ref=torch.where(v>0,v,v*torch.tensor(0.2,dtype=torch.float32))*torch.tensor(2**0.5,dtype=torch.float32)
print('unfused vs cpu-emul',float((yu.cpu()-ref).abs().max()),' fused vs cpu-emul',float((yf.cpu()-ref).abs().max()))
