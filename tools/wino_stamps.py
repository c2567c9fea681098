#!/usr/bin/env python3
"""Reads the in-kernel cycle stamps of a -DWINO_STAMP=1 build (diagnostic)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import ops, lib
B, h = 64, 128
S2 = os.environ.get("LAYER", "") == "s2"   # LAYER=s2: a 5x5/s2 layer (3x3 over space-to-depth, 16 chunks, zero-position skipping)
if S2:
    x = torch.randn(B, h, h, 512, device="cuda")
    w = ops.pack_wino_s2_weight(torch.randn(128, 128, 5, 5, device="cuda") * 0.05)
else:
    x = torch.randn(B, h, h, 128, device="cuda")
    w = ops.pack_wino_weight(torch.randn(128, 128, 3, 3, device="cuda") * 0.05)
bias = torch.randn(128, device="cuda"); beta = torch.rand(128, device="cuda") + 0.5; gamma = torch.rand(128, device="cuda") * 0.2
for _ in range(3):
    ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma, s2d_in=S2)
torch.cuda.synchronize()
L = lib.load()
buf = np.zeros(256 * 32, dtype=np.int64)
L.dsic_debug_wino_stamps.restype = ctypes.c_int
assert L.dsic_debug_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(256, 32).astype(np.float64)
d = lambda a, b: np.median(s[:, b] - s[:, a])
for c in range(4):
    print(f"chunk {c}: mfma {d(4*c, 4*c+1):8.0f}  barrier {d(4*c+2, 4*c+3):8.0f}")
    if c < 3: print(f"          gap to next chunk {d(4*c+3, 4*c+4):8.0f}")
print(f"fold + own stores -> E1 {d(24, 25):8.0f}   partner row added {d(25, 26):8.0f}   E2 {d(26, 27):8.0f}")
print(f"MFMA-wave tile total (chunk0 start -> after E2) {d(0, 27):8.0f}")
order = list(range(16)) + [24, 25, 26, 27]
print("raw deltas:", " ".join(f"{a}->{b}:{d(a, b):.0f}" for a, b in zip(order[:-1], order[1:])))
