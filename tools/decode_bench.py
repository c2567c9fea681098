#!/usr/bin/env python3
"""Timing of custom_compress / custom_decompress at the bench shape (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import entropy, synthetic as S
from dsic_amd.model import CompressionModel
B = int(os.environ.get("B", "64"))
m = CompressionModel(min_nu=2).cuda().eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.make_state_dict(seed=1).items()})
x = torch.from_numpy(S.make_patches(0, B, 256, 256)).cuda()
c = entropy.custom_compress(m, x)
for name, f in (("custom_compress (incl. forward + D2H)", lambda: entropy.custom_compress(m, x)),
                ("custom_decompress (incl. H2D)", lambda: entropy.custom_decompress(m, c))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.1f} ms per batch of {B} -> {B/dt:.0f} patches/s")
