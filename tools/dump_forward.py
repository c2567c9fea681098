#!/usr/bin/env python3
"""One forward pass of the synthetic model over bench patches, dumped for comparison between kernel variants
(tests/test_gpu_bench_parity.py runs it as a child with DSIC_WINO_BF16=0: the variant is read at import).
    python tools/dump_forward.py OUT.npz --size 256 --batch 64 --channels 3 --first 0"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsic_amd import metrics, synthetic as S
from dsic_amd.model import CompressionModel

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--first", type=int, default=0)
a = ap.parse_args()
B, H, W, C = a.batch, a.size, a.size, a.channels
m = CompressionModel(N=128, M=192, spatial_params=False, min_nu=2, max_nu=100.0, in_ch=C)
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.make_state_dict(seed=S.WEIGHT_SEED, in_ch=C).items()}, strict=True)
m = m.cuda().eval()
x = torch.from_numpy(S.make_patches(a.first, B, H, W, C)).cuda()
out = m(x, quant_mode="round")
bpp = (out.sums.sum(dim=1) / float(H * W)).double().cpu().numpy()
ms = metrics.ms_ssim_per_image(out["x_hat"], x, clamp_x=True).cpu().numpy()
np.savez_compressed(a.out, y_tilde=out["y_tilde"].cpu().numpy().astype(np.int16), z_tilde=out["z_tilde"].cpu().numpy().astype(np.int16),
                    bpp=bpp, msssim=ms, variant=np.array([int(os.environ.get("DSIC_WINO_BF16", "1"))]))
print(f"dumped {a.out}: mean bpp {bpp.mean():.6f} mean ms-ssim {ms.mean():.6f}")
