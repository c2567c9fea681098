"""Evaluation metrics of the reference harness on the GPU.

Mirrors the helpers of code/modelv2/modelseval.py:57-109 (pad, bpp, MSE, PSNR,
MS-SSIM with SSIM fallback) and the pytorch_msssim entry points they call
(`MS_SSIM`, `ms_ssim`, `ssim`).  pytorch-msssim 1.0.0 is third-party and absent
here; its algorithm is restated in csrc/metrics.hip.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import lib as _lib
from .ops import _f32c, _p, _stream

DEFAULT_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def pad_to_multiple_tensor(x, multiple=16):
    """modelseval.py:57-64: reflect-pad bottom/right so H and W are multiples of `multiple`.
    Returns (x_padded, pad_h, pad_w)."""
    x = _f32c(x, "pad_to_multiple_tensor")
    B, C, h, w = x.shape
    pad_h = (multiple - h % multiple) % multiple
    pad_w = (multiple - w % multiple) % multiple
    if pad_h == 0 and pad_w == 0:
        return x, 0, 0
    out = torch.empty((B, C, h + pad_h, w + pad_w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_reflect_pad_br(_p(x), _p(out), B * C, h, w, pad_h, pad_w, _stream()),
               "reflect_pad_br")
    return out, pad_h, pad_w


def _levels(X, Y, n_levels, data_range, clamp_x):
    """-> means [levels, B*C, 2] (mean cs, mean ssim) as a device fp64 tensor."""
    X = _f32c(X, "ms_ssim")
    Y = _f32c(Y, "ms_ssim")
    if X.shape != Y.shape or X.dim() != 4:
        raise ValueError(f"Input images should have the same 4-d shape, got {tuple(X.shape)} and {tuple(Y.shape)}")
    L = _lib.load()
    B, C, H, W = X.shape
    planes = B * C
    means = torch.empty((n_levels, planes, 2), dtype=torch.float64, device=X.device)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    x, y, clamp = X, Y, int(bool(clamp_x))
    for lvl in range(n_levels):
        h, w = x.shape[-2], x.shape[-1]
        partial = torch.empty(L.dsic_ssim_partial_doubles(planes, h, w), dtype=torch.float64, device=X.device)
        if lvl < n_levels - 1:
            # the level and the 2x2 average pool that feeds the next one, in one pass over x and y
            ho, wo = (h + 2 * (h % 2) - 2) // 2 + 1, (w + 2 * (w % 2) - 2) // 2 + 1
            nx = torch.empty((B, C, ho, wo), dtype=torch.float32, device=X.device)
            ny = torch.empty_like(nx)
            _lib.check(L.dsic_ssim_level_pool(_p(x), _p(y), _p(partial), _p(means[lvl]), _p(nx), _p(ny), planes, h, w,
                                              C1, C2, clamp, _stream()), "ssim_level_pool")
            x, y, clamp = nx, ny, 0
        else:
            _lib.check(L.dsic_ssim_level(_p(x), _p(y), _p(partial), _p(means[lvl]), planes, h, w, C1, C2, clamp,
                                         _stream()), "ssim_level")
    return means


_weight_cache = {}


def _weights_on(device, weights):
    key = (str(device), weights)
    w = _weight_cache.get(key)
    if w is None:
        w = torch.tensor(list(weights), dtype=torch.float32, device=device)
        _weight_cache[key] = w
    return w


def ms_ssim_per_image(X, Y, data_range=1.0, weights=(0.3, 0.5, 0.2), clamp_x=False):
    """Per-image MS-SSIM [B] (pytorch_msssim.ms_ssim(..., size_average=False))."""
    B, C, H, W = X.shape
    # pytorch-msssim 1.0.0 asserts this whatever len(weights) is
    assert min(H, W) > (11 - 1) * 2 ** 4, \
        "Image size should be larger than %d due to the 4 downsamplings in ms-ssim" % ((11 - 1) * 2 ** 4)
    w = _weights_on(X.device, tuple(float(v) for v in weights))   # cached: a host->device copy per call stalls the stream
    means = _levels(X, Y, len(weights), data_range, clamp_x)
    out = torch.empty(B, dtype=torch.float32, device=X.device)
    _lib.check(_lib.load().dsic_msssim_finalize(_p(means), _p(w), _p(out), len(weights), B, C, 1, _stream()),
               "msssim_finalize")
    return out


def ms_ssim(X, Y, data_range=255, size_average=True, weights=None, clamp_x=False):
    """pytorch_msssim.ms_ssim (eval_selfcontained_entropy.py:154)."""
    v = ms_ssim_per_image(X, Y, data_range, DEFAULT_WEIGHTS if weights is None else weights, clamp_x)
    return v.mean() if size_average else v


def ssim(X, Y, data_range=255, size_average=True, clamp_x=False):
    """pytorch_msssim.ssim — the fallback at modelseval.py:87-88 (no relu)."""
    B, C = X.shape[:2]
    means = _levels(X, Y, 1, data_range, clamp_x)
    w = torch.ones(1, dtype=torch.float32, device=X.device)
    out = torch.empty(B, dtype=torch.float32, device=X.device)
    _lib.check(_lib.load().dsic_msssim_finalize(_p(means), _p(w), _p(out), 1, B, C, 0, _stream()),
               "msssim_finalize")
    return out.mean() if size_average else out


class MS_SSIM(torch.nn.Module):
    """pytorch_msssim.MS_SSIM as constructed at modelseval.py:80-85."""

    def __init__(self, data_range=255, size_average=True, channel=3, weights=None):
        super().__init__()
        self.data_range, self.size_average, self.channel = data_range, size_average, channel
        self.weights = weights

    def forward(self, X, Y):
        return ms_ssim(X, Y, self.data_range, self.size_average, self.weights)


def sqerr_per_image(a, b, clamp_a=False):
    a = _f32c(a, "mse")
    b = _f32c(b, "mse")
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    B = a.shape[0]
    out = torch.empty(B, dtype=torch.float64, device=a.device)
    _lib.check(_lib.load().dsic_sqerr_per_image(_p(a), _p(b), _p(out), B, a[0].numel(), int(bool(clamp_a)),
                                                _stream()), "sqerr")
    return out


def mse(a, b):
    """F.mse_loss(a, b) over the whole batch (model.py:82, modelseval.py:69-70)."""
    return (sqerr_per_image(a, b).sum() / a.numel()).float()


def mse_per_image(a, b, clamp_a=False):
    return sqerr_per_image(a, b, clamp_a) / a[0].numel()


def psnr_from_mse(mse_value, max_val=1.0):
    """modelseval.py:72-76."""
    m = float(mse_value)
    return float("inf") if m == 0 else 10.0 * math.log10((max_val * max_val) / m)


def compute_bpp_from_out(out, orig_pixels):
    """modelseval.py:90-94: un-clamped (sum nll_y + sum nll_z) / un-padded pixels."""
    sums = getattr(out, "sums", None)
    if sums is not None:
        return float(sums.sum().item()) / float(orig_pixels)
    return float(out["nll_y"].double().sum().item() + out["nll_z"].double().sum().item()) / float(orig_pixels)
