#!/usr/bin/env python3
"""Reads the in-kernel cycle stamps of a -DWB_STAMP=1 build of conv_wino_bf16.hip (diagnostic).
LAYER=s2: conv(128,128,5,2) over space-to-depth (32 chunks); default conv(128,128,3,1) (8 chunks)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import ops, lib
B, h = 64, 128
S2 = os.environ.get("LAYER", "") == "s2"
if S2:
    x = torch.randn(B, h, h, 512, device="cuda")
    w = ops.split_wino_weight_bf16(ops.pack_wino_s2_weight(torch.randn(128, 128, 5, 5, device="cuda") * 0.05), 128, 512)
    n = 32
else:
    x = torch.randn(B, h, h, 128, device="cuda")
    w = ops.split_wino_weight_bf16(ops.pack_wino_weight(torch.randn(128, 128, 3, 3, device="cuda") * 0.05), 128, 128)
    n = 8
bias = torch.randn(128, device="cuda"); beta = torch.rand(128, device="cuda") + 0.5; gamma = torch.rand(128, device="cuda") * 0.2
for _ in range(3):
    ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma, s2d_in=S2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma, s2d_in=S2); e1.record(); torch.cuda.synchronize()
print("kernel ms", e0.elapsed_time(e1))
L = lib.load()
buf = np.zeros(256 * 128, dtype=np.int64)
L.dsic_debug_wb_stamps.restype = ctypes.c_int
assert L.dsic_debug_wb_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(256, 128).astype(np.float64)
m, hlp = s[:, :64], s[:, 64:]
d = lambda arr, a, b: np.median(arr[:, b] - arr[:, a])
C0 = int(os.environ.get("C0", "0"))          # build with -DWB_STAMP_C0=<C0> to see the chunks from C0 on
nc = min(n - C0, 20)
print("MFMA wave 0:  chunk: mfma-phase | barrier wait | gap")
for c in range(nc):
    print(f"  {c + C0:2d}: {d(m, 3*c, 3*c+1):7.0f} | {d(m, 3*c+1, 3*c+2):7.0f} | {d(m, 3*c+2, 3*c+3) if c + 1 < nc else 0:7.0f}")
print(f"  fold->E1 {d(m, 60, 61):7.0f}  finish {d(m, 61, 62):7.0f}  E2 wait {d(m, 62, 63):7.0f}")
print("helper wave 8: phase: stage (waits for the window loads) | commit | barrier wait | issue+stores+gap")
for c in range(min(nc, 15)):
    print(f"  {c + C0:2d}: {d(hlp, 4*c, 4*c+1):7.0f} | {d(hlp, 4*c+1, 4*c+2):7.0f} | {d(hlp, 4*c+2, 4*c+3):7.0f} | {d(hlp, 4*c+3, 4*c+4) if c + 1 < min(nc, 15) else 0:7.0f}")
print(f"per-chunk period (MFMA wave): {d(m, 0, 3*(nc-1)) / (nc-1):.0f} cycles")
