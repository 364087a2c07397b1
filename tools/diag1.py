import torch, sys
sys.path.insert(0,'.')
import ideas_amd.op as op
import torch.nn.functional as F
torch.manual_seed(0)
x=torch.randn(3,8,5,7); b=torch.randn(8)
ref=F.leaky_relu(x+b.view(1,-1,1,1),0.2)*(2**0.5)
for cl in (False,True):
    xd=x.cuda()
    if cl: xd=xd.contiguous(memory_format=torch.channels_last)
    y=op.fused_leaky_relu(xd,b.cuda())
    d=(y.cpu()-ref).abs()
    print(cl, float(d.max()), int((d>0).sum()), d.numel())
    idx=d.flatten().argmax()
    print(' at', float(y.cpu().flatten()[idx]), float(ref.flatten()[idx]), float((x+b.view(1,-1,1,1)).flatten()[idx]))
# gpu torch reference
refg=(F.leaky_relu(x.cuda()+b.cuda().view(1,-1,1,1),0.2)*(2**0.5)).cpu()
print('torch-gpu vs torch-cpu', float((refg-ref).abs().max()))
v=(x+b.view(1,-1,1,1))
alt=torch.where(v>0, v*(2**0.5), v*(0.2*2**0.5))
print('alt assoc vs ref', float((alt-ref).abs().max()))
y=op.fused_leaky_relu(x.cuda(),b.cuda())
print('ours vs alt', float((y.cpu()-alt).abs().max()))
alt2=torch.where(v>0, v, v*0.2)*torch.tensor(2**0.5,dtype=torch.float32)
print('ours vs alt2', float((y.cpu()-alt2).abs().max()))
