"""The forward operators as PyTorch custom ops (`torch.ops.dsic.*`), thin wrappers over the C ABI.

north_star / SURVEY.md section 8b: "driven through PyTorch-ROCm custom ops".  The dispatcher supplies what the
raw ctypes calls of ops.py leave to the caller: an entry in the operator registry (schema, `torch.ops.dsic.*`),
device-type dispatch (a CPU tensor is refused by the dispatcher, there is no CPU kernel), fake-tensor shape rules for
tracing tools, and the current-stream semantics of an ordinary CUDA op (the wrappers enqueue on
torch.cuda.current_stream()).  The kernels, the packing and the error text are those of ops.py / libdsic_hip.so; the
model's own forward keeps the direct calls (one Python frame less per launch).

    dsic::conv2d_bias_act    layers.py:29-31 conv() + fused GDN / IGDN / ReLU          (ops.conv2d_nhwc)
    dsic::conv3x3_wino       the same for 3x3 / 5x5-s2 layers on the Winograd kernels  (ops.conv3x3_wino_nhwc)
    dsic::convT5s2_bias_act  layers.py:83-97 ConvTranspose2d(.,.,5,2,2,1) + activation (ops.conv_transpose2d_wino_nhwc)
    dsic::rate               model.py:44-59 round + Student-t / Gaussian bits + sums   (ops.rate)
    dsic::range_encode       eval_selfcontained_entropy.py:36-62 support, tables, coder (entropy.compress_latents)
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor

from . import entropy as _entropy
from . import ops as _ops

_DEV = "cuda"


@torch.library.custom_op("dsic::conv2d_bias_act", mutates_args=(), device_types=_DEV)
def conv2d_bias_act(x: Tensor, w_packed: Tensor, bias: Tensor, beta: Optional[Tensor], gamma: Optional[Tensor],
                    Cout: int, k: int, stride: int, act: int) -> Tensor:
    return _ops.conv2d_nhwc(x, w_packed, bias, Cout, k, stride, act, beta, gamma)


@conv2d_bias_act.register_fake
def _(x, w_packed, bias, beta, gamma, Cout, k, stride, act):
    B, H, W, _ = x.shape
    return x.new_empty((B, (H + stride - 1) // stride, (W + stride - 1) // stride, Cout))


@torch.library.custom_op("dsic::conv3x3_wino", mutates_args=(), device_types=_DEV)
def conv3x3_wino(x: Tensor, u_packed: Tensor, bias: Tensor, beta: Optional[Tensor], gamma: Optional[Tensor],
                 Cout: int, act: int, s2d_in: bool, s2d_out: bool) -> Tensor:
    return _ops.conv3x3_wino_nhwc(x, u_packed, bias, Cout, act, beta, gamma, s2d_out=s2d_out, s2d_in=s2d_in)


@conv3x3_wino.register_fake
def _(x, u_packed, bias, beta, gamma, Cout, act, s2d_in, s2d_out):
    B, H, W, _ = x.shape
    return x.new_empty((B, H // 2, W // 2, 4 * Cout) if s2d_out else (B, H, W, Cout))


@torch.library.custom_op("dsic::convT5s2_bias_act", mutates_args=(), device_types=_DEV)
def convT5s2_bias_act(x: Tensor, u_packed4: Tensor, bias: Tensor, beta: Optional[Tensor], gamma: Optional[Tensor],
                      Cout: int, act: int) -> Tensor:
    return _ops.conv_transpose2d_wino_nhwc(x, u_packed4, bias, Cout, act, beta, gamma)


@convT5s2_bias_act.register_fake
def _(x, u_packed4, bias, beta, gamma, Cout, act):
    B, H, W, _ = x.shape
    return x.new_empty((B, 2 * H, 2 * W, Cout))


@torch.library.custom_op("dsic::rate", mutates_args=(), device_types=_DEV)
def rate(y_nhwc: Tensor, z_nhwc: Tensor, sigma: Tensor, nu: Tensor, z_log_sigma: Tensor) -> List[Tensor]:
    """-> [y_hat (NHWC), y_tilde, z_tilde, nll_y, nll_z (NCHW), sums [B,2] float64]"""
    r = _ops.rate(y_nhwc, z_nhwc, sigma, nu, z_log_sigma)
    return [r["y_hat_nhwc"], r["y_tilde"], r["z_tilde"], r["nll_y"], r["nll_z"], r["sums"]]


@rate.register_fake
def _(y_nhwc, z_nhwc, sigma, nu, z_log_sigma):
    B, Hy, Wy, M = y_nhwc.shape
    _, Hz, Wz, N = z_nhwc.shape
    yt, zt = y_nhwc.new_empty((B, M, Hy, Wy)), z_nhwc.new_empty((B, N, Hz, Wz))
    return [torch.empty_like(y_nhwc), yt, zt, torch.empty_like(yt), torch.empty_like(zt),
            y_nhwc.new_empty((B, 2), dtype=torch.float64)]


@torch.library.custom_op("dsic::range_encode", mutates_args=(), device_types=_DEV)
def range_encode(y_tilde: Tensor, z_tilde: Tensor, sigma_y: Tensor, nu_y: Tensor, sigma_z: Tensor, tail: int,
                 Lmax: int) -> List[Tensor]:
    """-> [bytes uint8 [B, cap_z + cap_y] (z string at 0, y string at cap_z), lengths int32 [B,2] (z, y),
    meta int32 [B,4] (min_y - tail, Ly, min_z - tail, Lz), err int32 [1]]"""
    c = _entropy.compress_latents(y_tilde, z_tilde, sigma_y, nu_y, sigma_z, tail, Lmax)
    return [c["bytes"], c["lengths"], c["meta"], c["err"].reshape(1)]


@range_encode.register_fake
def _(y_tilde, z_tilde, sigma_y, nu_y, sigma_z, tail, Lmax):
    B = y_tilde.shape[0]
    cap = _entropy._cap(y_tilde[0].numel()) + _entropy._cap(z_tilde[0].numel())
    return [y_tilde.new_empty((B, cap), dtype=torch.uint8), y_tilde.new_empty((B, 2), dtype=torch.int32),
            y_tilde.new_empty((B, 4), dtype=torch.int32), y_tilde.new_empty((1,), dtype=torch.int32)]
