#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the bench shapes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import ops

B = int(os.environ.get("B", "64"))
LAYERS = [  # name, kind, Cin, Cout, H(in), k, s
    ("g_a.0", "conv", 8, 128, 256, 3, 1),
    ("g_a.2", "conv", 128, 128, 256, 5, 2),
    ("g_a.4", "conv", 128, 128, 128, 3, 1),
    ("g_a.6", "conv", 128, 128, 128, 5, 2),
    ("g_a.8", "conv", 128, 128, 64, 3, 1),
    ("g_a.10", "conv", 128, 128, 64, 5, 2),
    ("g_a.12", "conv", 128, 128, 32, 3, 1),
    ("g_a.14", "conv", 128, 192, 32, 5, 2),
    ("g_s.0", "convT", 192, 128, 16, 5, 2),
    ("g_s.4", "convT", 128, 128, 32, 5, 2),
    ("g_s.8", "convT", 128, 128, 64, 5, 2),
    ("g_s.12", "img", 128, 3, 128, 5, 2),
]
only = os.environ.get("ONLY")
tot = 0.0
for name, kind, ci, co, h, k, s in LAYERS:
    if only and name not in only.split(","):
        continue
    x = torch.randn(B, h, h, ci, device="cuda")
    bias = torch.randn(co, device="cuda")
    beta = torch.rand(co, device="cuda") + 0.5
    gamma = torch.rand(co, device="cuda") * 0.2
    if name == "g_a.0":
        xi = torch.rand(B, 3, h, h, device="cuda")
        wr = torch.randn(co, 3, 3, 3, device="cuda") * 0.2
        f = lambda: ops.conv_first_nchw(xi, wr, bias, ops.ACT_GDN, beta, gamma)
        flops = 2.0 * B * h * h * co * 27
    elif kind == "conv" and k == 5 and co <= 128 and os.environ.get("DSIC_WINOGRAD", "1") != "0":
        xs = torch.randn(B, h // 2, h // 2, 4 * ci, device="cuda")
        w = ops.pack_wino_s2_weight(torch.randn(co, ci, 5, 5, device="cuda") * 0.05)
        f = lambda: ops.conv3x3_wino_nhwc(xs, w, bias, co, ops.ACT_GDN, beta, gamma, s2d_in=True)
        flops = 2.0 * B * (h // 2) ** 2 * co * ci * 25
    elif kind == "conv" and k == 3 and os.environ.get("DSIC_WINOGRAD", "1") != "0":
        w = ops.pack_wino_weight(torch.randn(co, ci, 3, 3, device="cuda") * 0.05)
        f = lambda: ops.conv3x3_wino_nhwc(x, w, bias, co, ops.ACT_GDN, beta, gamma)
        flops = 2.0 * B * h * h * co * ci * 9
    elif kind == "conv":
        w = ops.pack_conv_weight(torch.randn(co, ci, k, k, device="cuda") * 0.05)
        f = lambda: ops.conv2d_nhwc(x, w, bias, co, k, s, ops.ACT_GDN, beta, gamma)
        cin_real = 3 if ci == 8 else ci
        flops = 2.0 * B * (h // s) ** 2 * co * cin_real * k * k
    elif kind == "convT" and os.environ.get("DSIC_WINOGRAD", "1") != "0":
        w = ops.pack_wino_convT_weight(torch.randn(ci, co, 5, 5, device="cuda") * 0.05)
        f = lambda: ops.conv_transpose2d_wino_nhwc(x, w, bias, co, ops.ACT_IGDN, beta, gamma)
        flops = 2.0 * B * h * h * co * ci * 25
    elif kind == "convT":
        w = ops.pack_convT_weight(torch.randn(ci, co, 5, 5, device="cuda") * 0.05)
        f = lambda: ops.conv_transpose2d_nhwc(x, w, bias, co, ops.ACT_IGDN, beta, gamma)
        flops = 2.0 * B * h * h * co * ci * 25
    else:
        w = ops.pack_convT_image_weight(torch.randn(ci, co, 5, 5, device="cuda") * 0.05)
        f = lambda: ops.conv_transpose2d_image(x, w, bias, co)
        flops = 2.0 * B * h * h * co * ci * 25
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    print(f"{name:7s} {kind:5s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s")
print(f"total {tot:.3f} ms")
