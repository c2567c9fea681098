"""Oracle metrics vs an independent float64 numpy MS-SSIM (CPU only)."""
import numpy as np
import torch

from dsic_amd import synthetic as S
from oracle import ref_metrics as RM


def _np_msssim(X, Y, weights):
    """Independent float64 implementation with explicit loops over taps."""
    X = X.astype(np.float64)
    Y = Y.astype(np.float64)
    c = np.arange(11) - 5
    g = np.exp(-(c ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()

    def filt(a):
        H, W = a.shape[-2:]
        t = sum(g[k] * a[..., k:H - 10 + k, :] for k in range(11))
        return sum(g[k] * t[..., :, k:W - 10 + k] for k in range(11))

    def pool(a):
        H, W = a.shape[-2:]
        ph, pw = H % 2, W % 2
        p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(ph, ph), (pw, pw)])
        Ho, Wo = (H + 2 * ph - 2) // 2 + 1, (W + 2 * pw - 2) // 2 + 1
        p = p[..., :2 * Ho, :2 * Wo]
        return 0.25 * (p[..., 0::2, 0::2] + p[..., 0::2, 1::2] + p[..., 1::2, 0::2] + p[..., 1::2, 1::2])

    C1, C2 = 0.01 ** 2, 0.03 ** 2
    vals = []
    for i, w in enumerate(weights):
        m1, m2 = filt(X), filt(Y)
        s1, s2, s12 = filt(X * X) - m1 * m1, filt(Y * Y) - m2 * m2, filt(X * Y) - m1 * m2
        cs = (2 * s12 + C2) / (s1 + s2 + C2)
        ss = (2 * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1) * cs
        if i < len(weights) - 1:
            vals.append(np.maximum(cs.mean(axis=(-2, -1)), 0) ** w)
            X, Y = pool(X), pool(Y)
        else:
            vals.append(np.maximum(ss.mean(axis=(-2, -1)), 0) ** w)
    return np.prod(np.stack(vals), axis=0).mean(axis=1)


def _pair(B, H, W, seed):
    x = S.make_patches(seed, B, H, W)
    noise = S.hash_uniform(x.size, 99, seed).reshape(x.shape) - 0.5
    y = np.clip(0.8 * x + 0.1 + 0.1 * noise, 0, 1).astype(np.float32)
    return x, y


def test_ms_ssim_matches_float64_numpy():
    for (H, W, wts) in ((256, 256, (0.3, 0.5, 0.2)), (176, 200, RM.DEFAULT_WEIGHTS), (165, 163, (0.3, 0.5, 0.2))):
        x, y = _pair(2, H, W, 3)
        got = RM.ms_ssim(torch.from_numpy(y), torch.from_numpy(x), data_range=1.0, size_average=False,
                         weights=wts).numpy()
        want = _np_msssim(y, x, wts)
        assert np.max(np.abs(got - want)) < 2e-5, (got, want)
    a = torch.from_numpy(x)
    assert abs(RM.ms_ssim(a, a, data_range=1.0, weights=(0.3, 0.5, 0.2)).item() - 1.0) < 1e-6


def test_small_images_assert_like_pytorch_msssim():
    x, y = _pair(1, 120, 120, 1)
    try:
        RM.ms_ssim(torch.from_numpy(x), torch.from_numpy(y), data_range=1.0, weights=(0.3, 0.5, 0.2))
        raise RuntimeError("expected AssertionError")
    except AssertionError:
        pass
    s = RM.ssim(torch.from_numpy(x), torch.from_numpy(y), data_range=1.0).item()
    assert 0.0 < s < 1.0


def test_pad_psnr_helpers():
    x = torch.arange(2 * 3 * 5 * 7, dtype=torch.float32).reshape(2, 3, 5, 7) / 210
    same, ph, pw = RM.pad_to_multiple_tensor(x[:, :, :4, :4], 4)
    assert same.shape == (2, 3, 4, 4) and (ph, pw) == (0, 0)
    p4, ph, pw = RM.pad_to_multiple_tensor(x, 4)
    assert p4.shape == (2, 3, 8, 8) and (ph, pw) == (3, 1)
    assert torch.equal(p4[:, :, 5, :7], x[:, :, 3, :])       # reflect (no edge repeat)
    assert RM.compute_psnr(x, x) == float("inf")
    assert abs(RM.compute_psnr(x, x + 0.1) - 20.0) < 1e-4
