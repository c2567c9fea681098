import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from dsic_amd import ops
torch.manual_seed(0)
Cin, Cimg, H, W = 16, 3, 4, 4
x = torch.randn(1, Cin, H, W); w = torch.randn(Cin, Cimg, 5, 5) * 0.1; b = torch.zeros(Cimg)
ref = torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
got = ops.conv_transpose2d_image(x.permute(0,2,3,1).contiguous().cuda(), ops.pack_convT_image_weight(w.cuda()), b.cuda(), Cimg).cpu()
print("max err", (got-ref).abs().max().item(), "ref max", ref.abs().max().item())
# per-channel-of-input test: only channel c nonzero
for c in range(Cin):
    xc = torch.zeros_like(x); xc[:, c] = x[:, c]
    r = torch.nn.functional.conv_transpose2d(xc, w, b, stride=2, padding=2, output_padding=1)
    g = ops.conv_transpose2d_image(xc.permute(0,2,3,1).contiguous().cuda(), ops.pack_convT_image_weight(w.cuda()), b.cuda(), Cimg).cpu()
    print(c, round((g-r).abs().max().item(), 5), round(r.abs().max().item(), 3), round(g.abs().max().item(),3))
