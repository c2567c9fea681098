#!/usr/bin/env python3
"""Reads the in-kernel cycle stamps of a -DWBM_STAMP=1 build of conv_wino_bf16m.hip (diagnostic; third tile of every
workgroup, medians over workgroups).  LAYER=3x3 (default, 8 chunks per pass) | s2 (32) | convT (8)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import ops, lib
B, h = 64, 128
layer = os.environ.get("LAYER", "3x3")
bias = torch.randn(128, device="cuda"); beta = torch.rand(128, device="cuda") + 0.5; gamma = torch.rand(128, device="cuda") * 0.2
if layer == "s2":
    x = torch.randn(B, h, h, 512, device="cuda")
    w = ops.split_wino_weight_bf16(ops.pack_wino_s2_weight(torch.randn(128, 128, 5, 5, device="cuda") * 0.05), 128, 512)
    n = 32; run = lambda: ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma, s2d_in=True)
elif layer == "convT":
    x = torch.randn(B, h // 2, h // 2, 128, device="cuda")
    w = ops.split_wino_weight_bf16(ops.pack_wino_convT_weight(torch.randn(128, 128, 5, 5, device="cuda") * 0.05), 128, 128, 4)
    n = 8; run = lambda: ops.conv_transpose2d_wino_nhwc(x, w, bias, 128, ops.ACT_IGDN, beta, gamma)
else:
    x = torch.randn(B, h, h, 128, device="cuda")
    w = ops.split_wino_weight_bf16(ops.pack_wino_weight(torch.randn(128, 128, 3, 3, device="cuda") * 0.05), 128, 128)
    n = 8; run = lambda: ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(layer, "kernel ms", e0.elapsed_time(e1))
L = lib.load()
buf = np.zeros(256 * 256, dtype=np.int64)
L.dsic_debug_wbm_stamps.restype = ctypes.c_int
assert L.dsic_debug_wbm_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(256, 256).astype(np.float64)
s = s[s[:, 0] > 0]
m, hlp = s[:, :128], s[:, 128:]
d = lambda arr, a, b: np.median(arr[:, b] - arr[:, a])
ns = min(2 * n, 38)
print("MFMA wave 0:  chunk-pass: mfma-phase | barrier wait | gap       helper wave 8: stage | commit | barrier wait | issue+gap")
for c in range(ns):
    hl = f"{d(hlp, 4*c, 4*c+1):6.0f} | {d(hlp, 4*c+1, 4*c+2):6.0f} | {d(hlp, 4*c+2, 4*c+3):6.0f} | {d(hlp, 4*c+3, 4*c+4) if 4*c+4 < 128 and c + 1 < 2 * n else 0:6.0f}" if 4 * c + 3 < 128 else ""
    gap = d(m, 3*c+2, 3*c+3) if c + 1 < ns and c + 1 != n else 0
    print(f"  {c:2d}{'B' if c >= n else 'A'}: {d(m, 3*c, 3*c+1):7.0f} | {d(m, 3*c+1, 3*c+2):7.0f} | {gap:7.0f}        {hl}")
if 2 * n <= 38:
    print(f"mid fold: compute+write m0 {d(m, 3*(n-1)+2, 122):6.0f} | M1 wait {d(m, 122, 123):6.0f} | read+init+M2+write m1 {d(m, 123, 124):6.0f} | M3 wait {d(m, 124, 125):6.0f} | read+init+M4 {d(m, 125, 126):6.0f}")
    print(f"final fold (no barrier): {d(m, 120, 121):6.0f}")
    print(f"tile: {d(m, 0, 121):.0f} cycles; chunk-pass period (pass A, MFMA wave): {d(m, 0, 3*(n-1)) / (n-1):.0f}; pass B: {d(m, 3*n, 3*(2*n-1)) / (n-1):.0f}")
else:
    print(f"chunk-pass period (pass A, first {ns} chunk-passes): {d(m, 0, 3*(ns-1)) / (ns-1):.0f}")
