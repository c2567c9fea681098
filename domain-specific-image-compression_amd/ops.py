"""Thin torch-tensor wrappers over the C ABI (device pointers + current stream).

PyTorch is plumbing here: it owns device memory and the HIP stream; every
function forwards raw pointers to libdsic_hip.so.
"""
from __future__ import annotations

import ctypes

import torch

from . import lib as _lib
from .lib import ACT_GDN, ACT_IGDN, ACT_NONE, ACT_RELU  # noqa: F401


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (no CPU fallback), got {t.device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight [Cout,Cin,k,k] -> [k*k][CinP/8][CoutP][8]."""
    w = _f32c(w, "pack_conv_weight")
    Cout, Cin, k, k2 = w.shape
    assert k == k2
    L = _lib.load()
    n = L.dsic_packed_conv_weight_floats(Cout, Cin, k)
    dst = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(L.dsic_pack_conv_weight(_p(w), _p(dst), Cout, Cin, k, _stream()), "pack_conv_weight")
    return dst


def pack_convT_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.ConvTranspose2d weight [Cin,Cout,5,5] -> four phase kernels, 25 taps."""
    w = _f32c(w, "pack_convT_weight")
    Cin, Cout, k, k2 = w.shape
    if (k, k2) != (5, 5):
        raise ValueError("pack_convT_weight: kernel must be 5x5")
    dst = torch.empty(25 * (Cin // 8) * round_up(Cout, 32) * 8, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().dsic_pack_convT_weight(_p(w), _p(dst), Cin, Cout, _stream()),
               "pack_convT_weight")
    return dst


def pack_convT_image_weight(w: torch.Tensor) -> torch.Tensor:
    w = _f32c(w, "pack_convT_image_weight")
    Cin, Cimg, k, k2 = w.shape
    if (k, k2) != (5, 5):
        raise ValueError("pack_convT_image_weight: kernel must be 5x5")
    dst = torch.empty(9 * (Cin // 8) * 32 * 8, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().dsic_pack_convT_image_weight(_p(w), _p(dst), Cin, Cimg, _stream()),
               "pack_convT_image_weight")
    return dst


def image_to_nhwc8(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x, "image_to_nhwc8")
    B, C, H, W = x.shape
    dst = torch.empty((B, H, W, 8), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_image_to_nhwc8(_p(x), _p(dst), B, C, H, W, _stream()), "image_to_nhwc8")
    return dst


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x, "nhwc_to_nchw")
    B, H, W, C = x.shape
    dst = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_nhwc_to_nchw(_p(x), _p(dst), B, H, W, C, _stream()), "nhwc_to_nchw")
    return dst


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x, "nchw_to_nhwc")
    B, C, H, W = x.shape
    dst = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_nchw_to_nhwc(_p(x), _p(dst), B, C, H, W, _stream()), "nchw_to_nhwc")
    return dst


def conv2d_nhwc(x, w_packed, bias, Cout, k, stride, act=ACT_NONE, beta=None, gamma=None, out=None):
    """conv() + fused activation on NHWC activations (layers.py:29-31)."""
    x = _f32c(x, "conv2d_nhwc")
    B, H, W, CinP = x.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_conv2d_nhwc(_p(x), _p(w_packed), _p(bias), _p(beta), _p(gamma), _p(out),
                                            B, H, W, CinP, Cout, k, stride, act, _stream()),
               "conv2d_nhwc")
    return out


def conv_transpose2d_nhwc(x, w_packed, bias, Cout, act=ACT_NONE, beta=None, gamma=None, out=None):
    """ConvTranspose2d(Cin,Cout,5,2,2,output_padding=1) + fused activation."""
    x = _f32c(x, "conv_transpose2d_nhwc")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_conv_transpose2d_nhwc(_p(x), _p(w_packed), _p(bias), _p(beta), _p(gamma),
                                                      _p(out), B, H, W, Cin, Cout, act, _stream()),
               "conv_transpose2d_nhwc")
    return out


def conv_transpose2d_image(x, w_packed, bias, Cimg, out=None):
    """Last synthesis layer: NHWC features -> NCHW image [B,Cimg,2H,2W]."""
    x = _f32c(x, "conv_transpose2d_image")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, Cimg, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_conv_transpose2d_image(_p(x), _p(w_packed), _p(bias), _p(out), B, H, W,
                                                       Cin, Cimg, _stream()), "conv_transpose2d_image")
    return out
