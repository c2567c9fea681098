#!/usr/bin/env python3
"""Latent flips and bpp deviation of the GPU path against the 9 reference fixtures (tests/golden/forward_*.npz).
DSIC_WINO_BF16=1 (default): split-bf16 Winograd kernels; =0: fp32-input MFMA kernels."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import synthetic as S
from dsic_amd.model import CompressionModel
print(f"DSIC_WINO_BF16={os.environ.get('DSIC_WINO_BF16', '1')}")
print(f"{'fixture':28s} {'B':>2s} {'y flips':>8s} {'z flips':>8s} {'max |dbpp|':>11s} {'max |dx_hat|':>12s}")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in sorted(glob.glob(os.path.join(root, "tests", "golden", "forward_*.npz"))):
    g = np.load(path)
    B, C, H, W, seed, first = [int(v) for v in g["meta"]]
    spatial = bool(int(g["spatial"][0]))
    sd = S.make_state_dict(seed=seed, in_ch=C, spatial_params=spatial)
    m = CompressionModel(N=128, M=192, spatial_params=spatial, min_nu=2, max_nu=100.0, in_ch=C)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(S.make_patches(first, B, H, W, C)).cuda()
    out = m(x, quant_mode="round")
    yf = int((out["y_tilde"].cpu().numpy() != g["y_tilde"].astype(np.float32)).sum())
    zf = int((out["z_tilde"].cpu().numpy() != g["z_tilde"].astype(np.float32)).sum())
    bpp = out.sums.sum(dim=1).cpu().numpy() / (H * W)
    ref = (g["sum_nll_y"] + g["sum_nll_z"]) / (H * W)
    dx = float(np.abs(out["x_hat"][:, :, :32, :32].cpu().numpy() - g["x_hat_crop"]).max())
    print(f"{os.path.basename(path)[8:-4]:28s} {B:2d} {yf:8d} {zf:8d} {float(np.abs(bpp - ref).max()):11.2e} {dx:12.2e}")
