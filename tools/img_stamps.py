#!/usr/bin/env python3
"""Phase stamps of convT_image_dma_kernel (library built with -DIMG_STAMP=1 in tools/_abl/lib_imgstamp.so, see
tools/run_img_stamps.sh): s_memtime deltas of waves 0 (group 0) and 4 (group 1) of workgroup 0 over phases 16..31."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import ops
B, h = 64, 128
x = torch.randn(B, h, h, 128, device="cuda")
wp = ops.pack_convT_image_weight(torch.randn(128, 3, 5, 5, device="cuda") * 0.05)
b3 = torch.randn(3, device="cuda")
for _ in range(3): out = ops.conv_transpose2d_image(x, wp, b3, 3)
torch.cuda.synchronize()
st = out.flatten()[:128].view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
print("phase | wave 0: output + requests | split or MFMA | wait | barrier || wave 4: the same      (even phases: group 0 splits, group 1 contracts)")
for ph in range(15):
    row = []
    for w in range(2):
        s = st[w * 64 + ph * 4: w * 64 + ph * 4 + 5]
        row.append([int((s[i + 1] - s[i]) & 0xFFFFFFFF) for i in range(4)])
    print(f"{ph + 16:5d} | " + " | ".join(f"{v:6d}" for v in row[0]) + "  ||  " + " | ".join(f"{v:6d}" for v in row[1]) + f"   period {sum(row[0])}")
