#!/usr/bin/env python3
"""Cycle stamps of the fused-role Winograd kernel (-DWB_STAMP=1 build, DSIC_WINO_FUSED=1): waves 0 (M then T) and
4 (T then M) of one tile.  LAYER=s2 | 3x3."""
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
L.dsic_debug_wbx_stamps.restype = ctypes.c_int
assert L.dsic_debug_wbx_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(256, 128).astype(np.float64)
a, b = s[:, :64], s[:, 64:]
d = lambda arr, i, j: np.median(arr[:, j] - arr[:, i])
nc = min(n, 20)
print("period: wave 0  M | T | barrier      wave 4  T | M | barrier")
for c in range(nc):
    nxt = 3 * c + 3 if c + 1 < nc else 3 * c + 2
    print(f"  {c:2d}: {d(a, 3*c, 3*c+2):7.0f} | {0:7.0f} | {d(a, 3*c+2, nxt):7.0f}      "
          f"{d(b, 3*c, 3*c+2):7.0f} | {0:7.0f} | {d(b, 3*c+2, nxt):7.0f}")
print(f"  wave 0: fold->E1 {d(a, 60, 61):7.0f}  finish {d(a, 61, 62):7.0f}  E2 wait {d(a, 62, 63):7.0f}")
print(f"per-period (wave 0): {d(a, 0, 3*(nc-1)) / (nc-1):.0f} cycles")
