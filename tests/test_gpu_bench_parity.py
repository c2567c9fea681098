"""The split-bf16 default against the exact fp32-MFMA kernels ON THE WORKLOADS THE NUMBERS ARE QUOTED ON (the 64
bench patches of config 2/3 and the 32 patches of a config-5 shard), and the fp32 path as a full model against
the reference fixtures.  The kernel variant is read when dsic_amd is imported (and by two kernels through getenv),
so the fp32 path runs in a child process (code/modelv2/model.py:27-35,62: what a flipped round() does to bpp)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from dsic_amd import metrics, synthetic as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def _default_forward(B, H, C, first):
    from dsic_amd.model import CompressionModel
    m = CompressionModel(N=128, M=192, spatial_params=False, min_nu=2, max_nu=100.0, in_ch=C)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in S.make_state_dict(seed=S.WEIGHT_SEED, in_ch=C).items()}, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(S.make_patches(first, B, H, H, C)).cuda()
    out = m(x, quant_mode="round")
    bpp = (out.sums.sum(dim=1) / float(H * H)).double().cpu().numpy()
    ms = metrics.ms_ssim_per_image(out["x_hat"], x, clamp_x=True).cpu().numpy()
    return out["y_tilde"].cpu().numpy().astype(np.int16), out["z_tilde"].cpu().numpy().astype(np.int16), bpp, ms


@pytest.mark.parametrize("B,H,C,first", [(64, 256, 3, 0),      # bench.py's batch (config 2/3/4: rank 0's shard)
                                         (32, 512, 4, 0)])     # config-5 shard
def test_split_bf16_default_vs_exact_fp32_on_the_bench_batch(tmp_path, B, H, C, first):
    from dsic_amd import layers
    if not layers.WINO_BF16:
        pytest.skip("this process already runs the fp32 kernels")
    ref_file = str(tmp_path / "fp32.npz")
    _child(["tools/dump_forward.py", ref_file, "--size", str(H), "--batch", str(B), "--channels", str(C), "--first", str(first)],
           {"DSIC_WINO_BF16": "0"})
    ref = np.load(ref_file)
    assert int(ref["variant"][0]) == 0
    y, z, bpp, ms = _default_forward(B, H, C, first)
    yflips = (y != ref["y_tilde"]).reshape(B, -1).sum(axis=1)
    zflips = (z != ref["z_tilde"]).reshape(B, -1).sum(axis=1)
    dbpp = np.abs(bpp - ref["bpp"])
    dms = np.abs(ms - ref["msssim"])
    hist = np.bincount(yflips, minlength=5)
    print(f"B={B} {H}x{H}x{C}: y flips per image histogram {hist.tolist()} (max {yflips.max()}), z flips {int(zflips.sum())}, "
          f"max |dbpp| {dbpp.max():.2e}, max |dMS-SSIM| {dms.max():.2e}")
    zi = np.nonzero(zflips)[0]
    print(f"  images with a z flip: {zi.tolist()}; their |dbpp| {dbpp[zi].tolist()} |dMS-SSIM| {dms[zi].tolist()}")
    # north_star: bpp and MS-SSIM within 1e-4 of the reference; a flipped latent moves bpp by ~2e-5
    assert dbpp.max() < 1e-4 and dms.max() < 1e-4, (dbpp.max(), dms.max())
    # <= 4 flipped y latents per 256x256 patch (49 152 latents; a 512x512x4 patch has four times as many), and at most
    # one flipped z latent per patch: sigma / nu are means over all positions of h_s, so one z flip moves a patch's
    # bpp by ~1e-6 (measured: 5 of 64 bench patches, |dbpp| <= 3.3e-6; 11 of 32 config-5 patches, <= 2.6e-5)
    assert yflips.max() <= 4 * (H * H) // 65536, yflips
    assert zflips.max() <= 1, zflips


def test_exact_fp32_path_matches_the_reference_fixtures_with_no_flip():
    """DSIC_WINO_BF16=0 as a full model (advertised in INTEGRATION.md): 0 latent flips on all 9 fixtures."""
    out = _child(["tools/fixture_parity.py"], {"DSIC_WINO_BF16": "0"})
    rows = [ln.split() for ln in out.splitlines() if len(ln.split()) == 6 and ln.split()[1].isdigit()]
    assert len(rows) == len(glob.glob(os.path.join(ROOT, "tests", "golden", "forward_*.npz"))), out
    for name, B, yf, zf, dbpp, dx in rows:
        assert int(yf) == 0 and int(zf) == 0, (name, yf, zf)
        assert float(dbpp) < 1e-5 and float(dx) < 1e-4, (name, dbpp, dx)
