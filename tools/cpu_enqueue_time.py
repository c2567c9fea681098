#!/usr/bin/env python3
"""Host-side enqueue time of one step (forward + bpp + MS-SSIM) against its GPU time (diagnostic).
Measured (round 1): 0.6 ms of Python/ctypes per step against 11 ms on the GPU - the host runs far ahead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import metrics, synthetic as S
from dsic_amd.model import CompressionModel
dev = torch.device("cuda", 0)
sd = S.make_state_dict(seed=S.WEIGHT_SEED)
model = CompressionModel(N=128, M=192).to(dev).eval()
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model = model.to(dev)
x = torch.from_numpy(S.make_patches(0, 64)).to(dev)
def step():
    out = model(x, quant_mode="round")
    bpp = out.sums.sum(dim=1) / 65536.0
    ms = metrics.ms_ssim_per_image(out["x_hat"], x, clamp_x=True)
    return bpp.sum(), ms.sum()
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); step(); ts.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("cpu enqueue per step ms: median", np.median(ts) * 1e3, "max", max(ts) * 1e3, " total enqueue", (t1 - t0) * 1e3, "ms; wall incl. sync", (t2 - t0) * 1e3, "ms")
