"""ORACLE (test infrastructure, not product code): evaluation metrics on the CPU.

Restates, in eager PyTorch:
  pad_to_multiple_tensor / compute_mse / compute_psnr / compute_bpp_from_out
      code/modelv2/modelseval.py:57-76, 90-94
  MS-SSIM / SSIM as provided by pytorch-msssim 1.0.0 (Requirements.txt:216),
      called at modelseval.py:78-88 and eval_selfcontained_entropy.py:154.

Parity status of the MS-SSIM part: UNPINNED.  pytorch-msssim is third-party,
not vendored under /root/reference and not installed; the reference holds no
test or fixture for it (its CSV "msssim" column needs the missing checkpoints
and dataset).  The restatement follows the package's published algorithm
(Wang et al. MS-SSIM with an 11-tap sigma=1.5 Gaussian window, valid
filtering, K=(0.01,0.03), 2x2 average pooling between scales) and is
cross-checked in tests against an independent float64 numpy implementation.
The other helpers are plain arithmetic pinned by inspection.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

DEFAULT_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def pad_to_multiple_tensor(x, multiple=16):
    """modelseval.py:57-64: reflect-pad bottom/right to a multiple."""
    _, _, h, w = x.shape
    pad_h = (multiple - h % multiple) % multiple
    pad_w = (multiple - w % multiple) % multiple
    if pad_h == 0 and pad_w == 0:
        return x, 0, 0
    return F.pad(x, (0, pad_w, 0, pad_h), mode="reflect"), pad_h, pad_w


def compute_mse(x, y):
    return float(F.mse_loss(x, y, reduction="mean").item())


def compute_psnr(x, y, max_val=1.0):
    mse = compute_mse(x, y)
    return float("inf") if mse == 0 else 10.0 * math.log10((max_val * max_val) / mse)


def compute_bpp_from_out(out, orig_pixels):
    return float(out["nll_y"].sum().item() + out["nll_z"].sum().item()) / float(orig_pixels)


def _gauss_1d(size=11, sigma=1.5, dtype=torch.float32):
    coords = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, g):
    C = x.shape[1]
    k = g.numel()
    out = x
    if out.shape[2] >= k:
        out = F.conv2d(out, g.view(1, 1, k, 1).repeat(C, 1, 1, 1), groups=C)
    if out.shape[3] >= k:
        out = F.conv2d(out, g.view(1, 1, 1, k).repeat(C, 1, 1, 1), groups=C)
    return out


def _ssim_cs(X, Y, data_range, g):
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _filter(X, g), _filter(Y, g)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter(X * X, g) - mu1_sq
    s2 = _filter(Y * Y, g) - mu2_sq
    s12 = _filter(X * Y, g) - mu12
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ssim(X, Y, data_range=255, size_average=True):
    g = _gauss_1d(dtype=X.dtype)
    s, _ = _ssim_cs(X, Y, data_range, g)
    return s.mean() if size_average else s.mean(1)


def ms_ssim(X, Y, data_range=255, size_average=True, weights=None):
    assert min(X.shape[-2:]) > (11 - 1) * 2 ** 4, "Image size should be larger than 160"
    w = torch.tensor(DEFAULT_WEIGHTS if weights is None else list(weights), dtype=X.dtype)
    g = _gauss_1d(dtype=X.dtype)
    levels = w.numel()
    vals = []
    for i in range(levels):
        s, cs = _ssim_cs(X, Y, data_range, g)
        if i < levels - 1:
            vals.append(torch.relu(cs))
            pad = [d % 2 for d in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    vals.append(torch.relu(s))
    stack = torch.stack(vals, dim=0)                      # [levels, B, C]
    v = torch.prod(stack ** w.view(-1, 1, 1), dim=0)      # [B, C]
    return v.mean() if size_average else v.mean(1)
