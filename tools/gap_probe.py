#!/usr/bin/env python3
"""A few forward passes with no HIP events anywhere, for a rocprofv3 kernel trace of the launch gaps
(tools/step_timeline.py reads the trace)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsic_amd import metrics, synthetic as S           # noqa: E402
from dsic_amd.model import CompressionModel            # noqa: E402

m = CompressionModel(N=128, M=192, spatial_params=False, min_nu=2, max_nu=100.0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.make_state_dict(seed=1).items()}, strict=True)
m = m.cuda().eval()
x = torch.from_numpy(S.make_patches(0, 64, 256, 256)).cuda()
for _ in range(6):
    out = m(x, quant_mode="round")
    ms = metrics.ms_ssim_per_image(out["x_hat"], x, clamp_x=True)
torch.cuda.synchronize()
print(float(ms.mean()))
