#!/usr/bin/env python3
"""Runs one Winograd layer (split-bf16 or fp32 kernel) a few times, for rocprofv3 counter passes.
LAYER=3x3 | s2 | convT | image (the last layer, convT_image.hip) ; KERNEL=bf16 | fp32 ; B, HW from the environment (defaults: bench shape of the largest layer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import ops
layer = os.environ.get("LAYER", "3x3"); kern = os.environ.get("KERNEL", "bf16")
B = int(os.environ.get("B", "64")); h = int(os.environ.get("HW", "128")); reps = int(os.environ.get("REPS", "5"))
bias = torch.randn(128, device="cuda"); beta = torch.rand(128, device="cuda") + 0.5; gamma = torch.rand(128, device="cuda") * 0.2
if layer == "s2":
    x = torch.randn(B, h, h, 512, device="cuda")
    w = ops.pack_wino_s2_weight(torch.randn(128, 128, 5, 5, device="cuda") * 0.05)
    if kern == "bf16": w = ops.split_wino_weight_bf16(w, 128, 512)
    run = lambda: ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma, s2d_in=True)
elif layer == "image":
    x = torch.randn(B, h, h, 128, device="cuda")
    wp = ops.pack_convT_image_weight(torch.randn(128, 3, 5, 5, device="cuda") * 0.05)
    b3 = torch.randn(3, device="cuda")
    run = lambda: ops.conv_transpose2d_image(x, wp, b3, 3)
elif layer == "convT":
    x = torch.randn(B, h // 2, h // 2, 128, device="cuda")
    w = ops.pack_wino_convT_weight(torch.randn(128, 128, 5, 5, device="cuda") * 0.05)
    if kern == "bf16": w = ops.split_wino_weight_bf16(w, 128, 128, 4)
    run = lambda: ops.conv_transpose2d_wino_nhwc(x, w, bias, 128, ops.ACT_IGDN, beta, gamma)
else:
    x = torch.randn(B, h, h, 128, device="cuda")
    w = ops.pack_wino_weight(torch.randn(128, 128, 3, 3, device="cuda") * 0.05)
    if kern == "bf16": w = ops.split_wino_weight_bf16(w, 128, 128)
    run = lambda: ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma)
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
print(f"{layer} {kern}: {e0.elapsed_time(e1) / reps:.4f} ms per launch")
