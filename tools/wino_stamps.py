#!/usr/bin/env python3
"""Reads the in-kernel cycle stamps of a -DWINO_STAMP=1 build (diagnostic)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import ops, lib
B, h = 64, 128
x = torch.randn(B, h, h, 128, device="cuda")
w = ops.pack_wino_weight(torch.randn(128, 128, 3, 3, device="cuda") * 0.05)
bias = torch.randn(128, device="cuda"); beta = torch.rand(128, device="cuda") + 0.5; gamma = torch.rand(128, device="cuda") * 0.2
for _ in range(3):
    ops.conv3x3_wino_nhwc(x, w, bias, 128, ops.ACT_GDN, beta, gamma)
torch.cuda.synchronize()
L = lib.load()
buf = np.zeros(256 * 32, dtype=np.int64)
L.dsic_debug_wino_stamps.restype = ctypes.c_int
assert L.dsic_debug_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(256, 32).astype(np.float64)
d = lambda a, b: np.median(s[:, b] - s[:, a])
for c in range(4):
    print(f"chunk {c}: mfma {d(4*c, 4*c+1):8.0f}  transform {d(4*c+1, 4*c+2):8.0f}  barrier {d(4*c+2, 4*c+3):8.0f}")
    if c < 3: print(f"          gap to next chunk {d(4*c+3, 4*c+4):8.0f}")
print(f"last chunk end -> epilogue start {d(15, 24):8.0f}")
print(f"inverse+exchange {d(24, 25):8.0f}  act+store {d(25, 26):8.0f}  final barrier {d(26, 27):8.0f}")
print(f"tile total (chunk0 start -> end) {d(0, 27):8.0f}")
