#!/usr/bin/env python3
"""Times metrics.ms_ssim_per_image on the bench shape (B x C x H x W, 3 scales).  DSIC_SSIM_RB overrides the band height."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import metrics
B = int(os.environ.get("B", "64")); C = int(os.environ.get("C", "3")); H = int(os.environ.get("HW", "256"))
x = torch.rand(B, C, H, H, device="cuda"); y = (x + 0.05 * torch.randn_like(x)).clamp(0, 1)
for _ in range(3): v = metrics.ms_ssim_per_image(y, x, 1.0, (0.3, 0.5, 0.2), clamp_x=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): v = metrics.ms_ssim_per_image(y, x, 1.0, (0.3, 0.5, 0.2), clamp_x=True)
e1.record(); torch.cuda.synchronize()
print(f"ms_ssim B={B} C={C} {H}x{H}: {e0.elapsed_time(e1) / 20:.4f} ms per call, mean {v.mean().item():.6f}")
