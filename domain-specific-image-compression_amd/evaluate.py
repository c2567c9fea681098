"""Callers of the hot path, mirroring the reference's evaluation harness on device tensors.

  evaluate_batch      per-image rows of modelseval.py:158-200 (pad -> forward(round) -> crop ->
                      bpp / MSE / PSNR / MS-SSIM with the SSIM fallback of :78-88)
  evaluate_image      eval_selfcontained_entropy.py:126-159 (estimated vs coded bpp, 5-scale MS-SSIM
                      of the decoded reconstruction)
File I/O, CSV and plots of the reference scripts are out of scope.
"""
from __future__ import annotations

import torch

from . import entropy, metrics
from .model import rate_distortion_loss


def combine_bands(bands, want_uint8=False):
    """code/combinebandsall.py:7-12,35-36 on the GPU: bands [B,nb,H,W] raw reflectances (any float
    scale, e.g. B04,B03,B02[,B08]) -> per-band min-max normalised float32 [B,nb,H,W] in [0,1]
    (the model input) and, optionally, the uint8 image the reference saves."""
    from . import lib as _lib
    from .ops import _f32c, _p, _stream
    b = _f32c(bands, "combine_bands")
    B, nb, H, W = b.shape
    out = torch.empty_like(b)
    u8 = torch.empty((B, nb, H, W), dtype=torch.uint8, device=b.device) if want_uint8 else None
    _lib.check(_lib.load().dsic_normalize_bands(_p(b), _p(out), _p(u8), B * nb, H * W, _stream()), "normalize_bands")
    return (out, u8) if want_uint8 else out


@torch.no_grad()
def evaluate_batch(model, x, weights=(0.3, 0.5, 0.2)):
    """x: [B,C,h,w] float32 in [0,1] on the GPU, any size >= 16 — or the decoded images as uint8
    [B,h,w,C] (modelseval.py:164 `tensor_from_pil`: to_tensor runs on the GPU, fused into the first
    layer when no padding is needed).  Returns a list of per-image dicts with the columns of the
    reference's per-image CSV (modelseval.py:187-199): bpp, mse, psnr, msssim."""
    x_u8 = None
    if x.dtype == torch.uint8:
        from . import ops
        x_u8, x = x, ops.to_tensor_u8(x)                                    # :66-67
    B, C, h, w = x.shape
    if x_u8 is not None and h % 16 == 0 and w % 16 == 0:
        out = model(x_u8, quant_mode="round")                               # image bytes straight into g_a.0
    else:
        x_pad, _, _ = metrics.pad_to_multiple_tensor(x, 16)                 # :170
        out = model(x_pad, quant_mode="round")                              # :173
    x_hat = out["x_hat"][:, :, :h, :w].contiguous()                         # :178 (clamp fused below)
    bpp = (out.sums.sum(dim=1) / float(h * w)).cpu().numpy()                # :181, un-padded pixel count
    mse = metrics.mse_per_image(x_hat, x, clamp_a=True).cpu().numpy()       # :184
    try:
        ms = metrics.ms_ssim_per_image(x_hat, x, data_range=1.0, weights=weights, clamp_x=True)
    except AssertionError:                                                  # :87-88 fallback for small images
        ms = metrics.ssim(x_hat, x, data_range=1.0, size_average=False, clamp_x=True)
    ms = ms.cpu().numpy()
    return [{"bpp": float(bpp[i]), "mse": float(mse[i]), "psnr": metrics.psnr_from_mse(mse[i]),
             "msssim": float(ms[i])} for i in range(B)]


@torch.no_grad()
def evaluate_image(model, x, tail=10):
    """eval_selfcontained_entropy.py:126-159 on a device tensor x [B,C,H,W] (H, W multiples of 16)."""
    out = model(x, quant_mode="round")
    _, R_est, D = rate_distortion_loss(out, x, lambda_rd=1.0, dist="msssim")        # :141-143
    compressed = entropy.custom_compress(model, x, tail=tail)                       # :147
    bpp_real = entropy.real_bpp(compressed, x.size(-2), x.size(-1)) / x.size(0)     # :148-149 (per image)
    x_hat = entropy.custom_decompress(model, compressed)                            # :153
    mss = metrics.ms_ssim(x_hat, x, data_range=1.0, size_average=True).item()       # :154 (5 scales)
    return {"bpp_est": float(R_est.item()), "D_msssim3": float(D.item()), "bpp_real": float(bpp_real),
            "ms_ssim5": mss, "x_hat": x_hat, "compressed": compressed}
