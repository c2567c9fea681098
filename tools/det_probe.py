import sys, os
sys.path.insert(0, os.getcwd())
import torch
from dsic_amd import ops
torch.manual_seed(0)
B,H,W,C=2,16,32,128
x=torch.randn(B,H,W,C,device="cuda")
w=torch.randn(C,C,3,3,device="cuda")*0.05
u=ops.split_wino_weight_bf16(ops.pack_wino_weight(w),C,C)
b=torch.randn(C,device="cuda"); beta=torch.rand(C,device="cuda")+0.5; gamma=torch.rand(C,device="cuda")*0.2
ref=torch.nn.functional.conv2d(x.permute(0,3,1,2),w,b,padding=1)
for act,name in ((ops.ACT_NONE,"none"),(ops.ACT_GDN,"gdn"),(ops.ACT_RELU,"relu")):
    r=ref
    if act==ops.ACT_GDN: r=ref/torch.sqrt(beta.view(1,-1,1,1)+gamma.view(1,-1,1,1)*ref*ref)
    if act==ops.ACT_RELU: r=torch.relu(ref)
    outs=[ops.conv3x3_wino_nhwc(x,u,b,C,act,beta,gamma).permute(0,3,1,2) for _ in range(4)]
    print(name,"err",[float((o-r).abs().max()) for o in outs],"equal",[bool(torch.equal(outs[0],o)) for o in outs])
    d=(outs[0]-r).abs()
    idx=torch.nonzero(d>1e-3)
    print("  bad count",idx.shape[0], idx[:6].tolist())
d = (outs[0] - r).abs().amax(dim=(0, 2, 3))
print("bad channels", torch.nonzero(d > 1e-3).flatten().tolist())
o = ops.conv3x3_wino_nhwc(x, u, b, C, ops.ACT_NONE, beta, gamma).permute(0, 3, 1, 2)
dd = (o - ref)
for c in torch.nonzero(dd.abs().amax(dim=(0, 2, 3)) > 1e-3).flatten().tolist():
    print(c, "delta min/max", float(dd[:, c].min()), float(dd[:, c].max()), "bias", float(b[c]))
