#!/usr/bin/env python3
"""Achievable HBM bandwidth for a 2.1 GB tensor (the size of the largest activation): fill / copy / sum.
Measured on MI355X (round 1): fill 6.9 TB/s, copy 4.9 TB/s (read+write), sum 4.0 TB/s."""
import torch, time
x = torch.empty(64*256*256*128, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, f in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("read(sum)", lambda: x.sum())):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = x.numel() * 4 / 1e9 * (2 if name == "copy" else 1)
    print(f"{name}: {ms:.3f} ms  {gb/ms:.2f} TB/s")
