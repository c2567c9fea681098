"""Thin torch-tensor wrappers over the C ABI (device pointers + current stream).

PyTorch is plumbing here: it owns device memory and the HIP stream; every
function forwards raw pointers to libdsic_hip.so.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import lib as _lib
from .lib import ACT_GDN, ACT_IGDN, ACT_NONE, ACT_RELU  # noqa: F401


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    # one process drives one GPU (torch.distributed, one rank per device): the current stream of the
    # current device.  Tensors on another device are rejected by _f32c's callers' pointer checks.
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (no CPU fallback), got {t.device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # launches go to the current stream of the current device (_stream): a tensor of another GPU would be
        # read through the wrong stream, and its kernel attributes are set per device
        raise RuntimeError(f"{name}: tensor on {t.device}, current device is cuda:{torch.cuda.current_device()} "
                           "(one process drives one GPU; wrap the call in torch.cuda.device(...))")
    return t.contiguous()


# DSIC_WINO_SPLITK=0: never share a tile's input channels between workgroups (dsic_wino_bf16_ksplit decides otherwise)
WINO_SPLITK = os.environ.get("DSIC_WINO_SPLITK", "1") != "0"

# Optional per-launch timer (bench.py): object with .record(tag, flops, launch)
# where launch() enqueues the kernel on the current stream.
_kernel_timer = None


def set_kernel_timer(timer):
    global _kernel_timer
    _kernel_timer = timer


def _conv_kernel_name(win, stride, Wo, Cin, CoutP):
    """Template instantiation conv_igemm.hip dispatches to (for profile matching)."""
    ck = (32 if Cin % 32 == 0 else 8) if win == 3 else (16 if Cin % 16 == 0 else 8)
    if Wo > 8:
        tile = (16, 8, 1)
    elif Wo > 4:
        tile = (8, 8, 2)
    else:
        tile = (4, 4, 8)
        if win == 5:
            ck = 8
    nt = CoutP // 32
    ntw, narrow = (1, 1) if nt == 1 else ((1, 0) if nt <= 4 else (2, 0))
    return f"conv_igemm_kernel<{win},{stride},{tile[0]},{tile[1]},{tile[2]},{ck},{ntw},{narrow}>"


def _timed(name, flops, launch, exec_flops=None):
    """flops: algorithmic (direct-convolution) FLOPs of the launch; exec_flops: FLOPs the MFMA
    pipe actually executes (smaller for Winograd, larger where channels are padded)."""
    if _kernel_timer is None:
        return launch()
    return _kernel_timer.record(name, flops, launch, exec_flops if exec_flops is not None else flops)


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
    if Cin % 16 or not 1 <= Cimg <= 4:
        raise ValueError("pack_convT_image_weight: Cin must be a multiple of 16 and Cimg in [1,4]")
    dst = torch.empty(_lib.load().dsic_convT_image_weight_floats(Cin), dtype=torch.float32, device=w.device)
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


def conv2d_nhwc(x, w_packed, bias, Cout, k, stride, act=ACT_NONE, beta=None, gamma=None, out=None,
                cin_real=None):
    """conv() + fused activation on NHWC activations (layers.py:29-31)."""
    x = _f32c(x, "conv2d_nhwc")
    B, H, W, CinP = x.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    L = _lib.load()
    _timed(_conv_kernel_name(k, stride, Wo, CinP, round_up(Cout, 32)),
           2.0 * B * Ho * Wo * Cout * (cin_real or CinP) * k * k,
           lambda: _lib.check(L.dsic_conv2d_nhwc(_p(x), _p(w_packed), _p(bias), _p(beta), _p(gamma),
                                                 _p(out), B, H, W, CinP, Cout, k, stride, act, _stream()),
                              "conv2d_nhwc"))
    return out


def pack_wino_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d 3x3 weight [Cout,Cin,3,3] -> Winograd-domain U = G g G^T, [16][Cin/8][CoutP][8]."""
    w = _f32c(w, "pack_wino_weight")
    Cout, Cin, k, k2 = w.shape
    if (k, k2) != (3, 3):
        raise ValueError("pack_wino_weight: kernel must be 3x3")
    L = _lib.load()
    dst = torch.empty(L.dsic_wino_weight_floats(Cout, Cin), dtype=torch.float32, device=w.device)
    _lib.check(L.dsic_pack_wino_weight(_p(w), _p(dst), Cout, Cin, _stream()), "pack_wino_weight")
    return dst


def pack_wino_s2_weight(w: torch.Tensor) -> torch.Tensor:
    """5x5 stride-2 weight [Cout,Cs,5,5] -> Winograd U of the equivalent 3x3 conv over 4*Cs s2d channels."""
    w = _f32c(w, "pack_wino_s2_weight")
    Cout, Cs, k, k2 = w.shape
    if (k, k2) != (5, 5):
        raise ValueError("pack_wino_s2_weight: kernel must be 5x5")
    L = _lib.load()
    dst = torch.empty(L.dsic_wino_weight_floats(Cout, 4 * Cs), dtype=torch.float32, device=w.device)
    _lib.check(L.dsic_pack_wino_s2_weight(_p(w), _p(dst), Cout, Cs, _stream()), "pack_wino_s2_weight")
    return dst


_tickets = {}


def _ticket(device):
    """16 zeroed bytes per (device, stream): the persistent Winograd kernel's tile ticket."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _tickets.get(key)
    if t is None:
        t = torch.zeros(2, dtype=torch.int64, device=device)
        _tickets[key] = t
    return t


def pack_wino_convT_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d weight [Cin,Cout,5,5] -> Winograd U of its four 3x3 sub-pixel phase convs."""
    w = _f32c(w, "pack_wino_convT_weight")
    Cin, Cout, k, k2 = w.shape
    if (k, k2) != (5, 5):
        raise ValueError("pack_wino_convT_weight: kernel must be 5x5")
    L = _lib.load()
    dst = torch.empty(4 * L.dsic_wino_weight_floats(Cout, Cin), dtype=torch.float32, device=w.device)
    _lib.check(L.dsic_pack_wino_convT_weight(_p(w), _p(dst), Cin, Cout, _stream()), "pack_wino_convT_weight")
    return dst


def split_wino_weight_bf16(u_f32: torch.Tensor, Cout: int, Cin: int, nphase: int = 1) -> torch.Tensor:
    """fp32 transformed weights (pack_wino_*_weight) -> bf16 planes for the *_wino_bf16 kernels (uint8 tensor)."""
    L = _lib.load()
    dst = torch.empty(nphase * L.dsic_wino_bf16_weight_bytes(Cout, Cin), dtype=torch.uint8, device=u_f32.device)
    _lib.check(L.dsic_split_wino_weight_bf16(_p(u_f32), _p(dst), Cout, Cin, nphase, _stream()), "split_wino_weight_bf16")
    return dst


def wino_bf16_planes() -> int:
    return int(_lib.load().dsic_wino_bf16_planes())


def conv_transpose2d_wino_nhwc(x, u_packed4, bias, Cout, act=ACT_NONE, beta=None, gamma=None, out=None):
    """ConvTranspose2d(Cin,Cout,5,2,2,1) + fused activation: four Winograd 3x3 phase convs.
    u_packed4: fp32 transformed weights (fp32 MFMA kernel) or the uint8 bf16 planes of
    split_wino_weight_bf16 (split-bf16 kernel)."""
    x = _f32c(x, "conv_transpose2d_wino_nhwc")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    L = _lib.load()
    wino_tiles = 4 * B * (-(-H // 8)) * (-(-W // 16)) * 32
    if u_packed4.dtype == torch.uint8:
        nprod = 3 if wino_bf16_planes() == 2 else 6
        # large layers run on the 64-tile two-pass kernel (conv_wino_bf16m.hip): the symbol the profiler will show
        m64 = bool(L.dsic_wino_bf16_m64(H, W, Cin, 4))
        _timed("conv_wino_bf16m_kernel<2>" if m64 else "conv_wino_bf16_kernel<2>", 2.0 * B * H * W * Cout * Cin * 25,
               lambda: _lib.check(L.dsic_conv_transpose2d_wino_bf16_nhwc(_p(x), _p(u_packed4), _p(bias), _p(beta),
                                                                         _p(gamma), _p(out), B, H, W, Cin, Cout, act,
                                                                         _p(_ticket(x.device)), _stream()),
                                  "conv_transpose2d_wino_bf16_nhwc"),
               exec_flops=2.0 * nprod * wino_tiles * 12.25 * Cin * round_up(Cout, 32))
        return out
    _timed("conv_wino_kernel<2>", 2.0 * B * H * W * Cout * Cin * 25,
           lambda: _lib.check(L.dsic_conv_transpose2d_wino_nhwc(_p(x), _p(u_packed4), _p(bias), _p(beta), _p(gamma),
                                                                _p(out), B, H, W, Cin, Cout, act, _p(_ticket(x.device)),
                                                                _stream()),
                              "conv_transpose2d_wino_nhwc"),
           exec_flops=2.0 * wino_tiles * 12.25 * Cin * round_up(Cout, 32))
    return out


def space_to_depth(x_nhwc):
    """[B,H,W,C] -> [B,H/2,W/2,4C] with channel (a*2+b)*C+c = x[2i+a][2j+b][c] (layout plumbing)."""
    B, H, W, C = x_nhwc.shape
    return x_nhwc.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * C).contiguous()


def depth_to_space(x_s2d):
    B, H2, W2, C4 = x_s2d.shape
    C = C4 // 4
    return x_s2d.view(B, H2, W2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H2, 2 * W2, C).contiguous()


def conv3x3_wino_nhwc(x, u_packed, bias, Cout, act=ACT_NONE, beta=None, gamma=None, out=None, s2d_out=False,
                      algo_flops=None, s2d_in=False, out_coff=0, split_k=True):
    """conv(Cin,Cout,3,1) + fused activation by Winograd F(2x2,3x3) on NHWC activations.

    s2d_out: write [B,H/2,W/2,4*Cout] (space-to-depth) for a following 5x5/s2 layer."""
    x = _f32c(x, "conv3x3_wino_nhwc")
    B, H, W, Cin = x.shape
    if out is None:
        shape = (B, H // 2, W // 2, 4 * Cout) if s2d_out else (B, H, W, Cout)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    L = _lib.load()
    wino_tiles = B * (-(-H // 8)) * (-(-W // 16)) * 32          # 2x2-output tiles incl. border padding
    if u_packed.dtype == torch.uint8:                            # bf16 planes: split-bf16 kernel
        nprod = 3 if wino_bf16_planes() == 2 else 6
        ksplit = L.dsic_wino_bf16_ksplit(H, W, Cin) if (WINO_SPLITK and split_k) else 1
        if ksplit > 1:
            # few tiles per image: the input channels of a tile are shared by `ksplit` work items
            partials = torch.empty((ksplit,) + tuple(out.shape), dtype=torch.float32, device=x.device)
            _timed("conv_wino_bf16_kernel<1>" if s2d_in else "conv_wino_bf16_kernel<0>",
                   algo_flops if algo_flops is not None else 2.0 * B * H * W * Cout * Cin * 9,
                   lambda: _lib.check(L.dsic_conv3x3_wino_bf16_splitk_nhwc(
                       _p(x), _p(u_packed), _p(bias), _p(beta), _p(gamma), _p(out), B, H, W, Cin, Cout, act,
                       int(bool(s2d_out)), int(bool(s2d_in)), 0 if s2d_out else int(out.shape[-1]), int(out_coff),
                       ksplit, _p(partials), _p(_ticket(x.device)), _stream()), "conv3x3_wino_bf16_splitk_nhwc"),
                   exec_flops=2.0 * nprod * wino_tiles * (12.25 if s2d_in else 16) * Cin * round_up(Cout, 32))
            return out
        m64 = "m" if L.dsic_wino_bf16_m64(H, W, Cin, 1) else ""   # 64-tile two-pass kernel (conv_wino_bf16m.hip)
        _timed(f"conv_wino_bf16{m64}_kernel<1>" if s2d_in else f"conv_wino_bf16{m64}_kernel<0>",
               algo_flops if algo_flops is not None else 2.0 * B * H * W * Cout * Cin * 9,
               lambda: _lib.check(L.dsic_conv3x3_wino_bf16_nhwc(_p(x), _p(u_packed), _p(bias), _p(beta), _p(gamma),
                                                                _p(out), B, H, W, Cin, Cout, act, int(bool(s2d_out)),
                                                                int(bool(s2d_in)),
                                                                0 if s2d_out else int(out.shape[-1]), int(out_coff),
                                                                _p(_ticket(x.device)), _stream()),
                                  "conv3x3_wino_bf16_nhwc"),
               exec_flops=2.0 * nprod * wino_tiles * (12.25 if s2d_in else 16) * Cin * round_up(Cout, 32))
        return out
    _timed("conv_wino_kernel<1>" if s2d_in else "conv_wino_kernel<0>",
           algo_flops if algo_flops is not None else 2.0 * B * H * W * Cout * Cin * 9,
           lambda: _lib.check(L.dsic_conv3x3_wino_nhwc(_p(x), _p(u_packed), _p(bias), _p(beta), _p(gamma), _p(out),
                                                       B, H, W, Cin, Cout, act, int(bool(s2d_out)), int(bool(s2d_in)),
                                                       _p(_ticket(x.device)), _stream()),
                              "conv3x3_wino_nhwc"),
           exec_flops=2.0 * wino_tiles * (12.25 if s2d_in else 16) * Cin * round_up(Cout, 32))
    return out


def to_tensor_u8(x_u8_nhwc: torch.Tensor) -> torch.Tensor:
    """torchvision to_tensor (modelseval.py:66-67): uint8 [B,H,W,C] on the GPU -> float32 [B,C,H,W] in [0,1]."""
    if not x_u8_nhwc.is_cuda or x_u8_nhwc.dtype != torch.uint8 or x_u8_nhwc.dim() != 4:
        raise TypeError("to_tensor_u8: expected a uint8 [B,H,W,C] tensor on the GPU")
    x = x_u8_nhwc.contiguous()
    B, H, W, C = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dsic_image_u8hwc_to_f32nchw(_p(x), _p(out), B, C, H, W, _stream()), "to_tensor_u8")
    return out


def conv_first_nchw(x, w, bias, act=ACT_NONE, beta=None, gamma=None, s2d_out=False):
    """conv(Cimg,Cout,3,1) + fused activation from the image to NHWC (layers.py:51).
    x: float32 NCHW in [0,1] (the reference's tensor contract), or uint8 NHWC image bytes
    (to_tensor fused into the kernel)."""
    w = _f32c(w, "conv_first_nchw")
    Cout = w.shape[0]
    if x.dtype == torch.uint8:
        if not x.is_cuda:
            raise RuntimeError(f"conv_first_nchw: expected a tensor on the GPU (no CPU fallback), got {x.device}")
        x = x.contiguous()
        B, H, W, C = x.shape
        shape = (B, H // 2, W // 2, 4 * Cout) if s2d_out else (B, H, W, Cout)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        L = _lib.load()
        _timed(f"conv_first_kernel<{C}>", 2.0 * B * H * W * Cout * C * 9,
               lambda: _lib.check(L.dsic_conv_first_u8hwc(_p(x), _p(w), _p(bias), _p(beta), _p(gamma), _p(out), B, C,
                                                          H, W, Cout, act, int(bool(s2d_out)), _stream()),
                                  "conv_first_u8hwc"))
        return out
    x = _f32c(x, "conv_first_nchw")
    B, C, H, W = x.shape
    shape = (B, H // 2, W // 2, 4 * Cout) if s2d_out else (B, H, W, Cout)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    L = _lib.load()
    _timed(f"conv_first_kernel<{C}>", 2.0 * B * H * W * Cout * C * 9,
           lambda: _lib.check(L.dsic_conv_first_nchw(_p(x), _p(w), _p(bias), _p(beta), _p(gamma), _p(out), B, C,
                                                     H, W, Cout, act, int(bool(s2d_out)), _stream()),
                              "conv_first_nchw"))
    return out


def conv_transpose2d_nhwc(x, w_packed, bias, Cout, act=ACT_NONE, beta=None, gamma=None, out=None):
    """ConvTranspose2d(Cin,Cout,5,2,2,output_padding=1) + fused activation."""
    x = _f32c(x, "conv_transpose2d_nhwc")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    L = _lib.load()
    _timed(_conv_kernel_name(3, 1, W, Cin, round_up(Cout, 32)),
           2.0 * B * H * W * Cout * Cin * 25,
           lambda: _lib.check(L.dsic_conv_transpose2d_nhwc(_p(x), _p(w_packed), _p(bias), _p(beta),
                                                           _p(gamma), _p(out), B, H, W, Cin, Cout, act,
                                                           _stream()), "conv_transpose2d_nhwc"))
    return out


def conv_transpose2d_image(x, w_packed, bias, Cimg, out=None):
    """Last synthesis layer: NHWC features -> NCHW image [B,Cimg,2H,2W]."""
    x = _f32c(x, "conv_transpose2d_image")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, Cimg, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    L = _lib.load()
    _timed("convT_image_kernel", 2.0 * B * H * W * Cimg * Cin * 25,
           lambda: _lib.check(L.dsic_conv_transpose2d_image(_p(x), _p(w_packed), _p(bias), _p(out), B, H, W,
                                                            Cin, Cimg, _stream()), "conv_transpose2d_image"),
           exec_flops=2.0 * B * (-(-H // 16) * 16) * (-(-W // 32) * 32) * 16 * Cin * 9)
    return out


def round_half_even(x: torch.Tensor) -> torch.Tensor:
    """quantize(x, "round") (model.py:32-33)."""
    x = _f32c(x, "round")
    out = torch.empty_like(x)
    _lib.check(_lib.load().dsic_round(_p(x), _p(out), x.numel(), _stream()), "round")
    return out


def gdn_nchw(x, beta_eff, gamma_eff, inverse: bool):
    """Stand-alone GDN / IGDN on NCHW (layers.py:19-27)."""
    x = _f32c(x, "gdn")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().dsic_gdn_nchw(_p(x), _p(beta_eff), _p(gamma_eff), _p(out), B, C, H * W,
                                         int(bool(inverse)), _stream()), "gdn")
    return out


def hyper_params(t_nhwc, w1s, b1s, w2s, b2s, w1n, b1n, w2n, b2n, M, min_nu, max_nu):
    """pool + MLP heads + exp/clamp -> (log_sigma, log_nu, sigma, nu), each [B,M]."""
    t = _f32c(t_nhwc, "hyper_params")
    B, H, W, N = t.shape
    outs = [torch.empty((B, M), dtype=torch.float32, device=t.device) for _ in range(4)]
    _lib.check(_lib.load().dsic_hyper_params(_p(t), _p(w1s), _p(b1s), _p(w2s), _p(b2s), _p(w1n), _p(b1n),
                                             _p(w2n), _p(b2n), _p(outs[0]), _p(outs[1]), _p(outs[2]),
                                             _p(outs[3]), B, H * W, N, M, float(min_nu), float(max_nu),
                                             _stream()), "hyper_params")
    return outs


def rate(y_nhwc, z_nhwc, sigma, nu, z_log_sigma, y_noisy=None, z_noisy=None):
    """round + Student-t/Gaussian bits + per-image sums.

    Returns dict(y_hat_nhwc, y_tilde, z_tilde, nll_y, nll_z (NCHW), sums [B,2] fp64).
    """
    y = _f32c(y_nhwc, "rate")
    z = _f32c(z_nhwc, "rate")
    B, Hy, Wy, M = y.shape
    _, Hz, Wz, N = z.shape
    dev = y.device
    y_hat = torch.empty_like(y)
    y_t = torch.empty((B, M, Hy, Wy), dtype=torch.float32, device=dev)
    nll_y = torch.empty_like(y_t)
    z_t = torch.empty((B, N, Hz, Wz), dtype=torch.float32, device=dev)
    nll_z = torch.empty_like(z_t)
    sums = torch.empty((B, 2), dtype=torch.float64, device=dev)
    work = torch.empty(_lib.load().dsic_rate_workspace_doubles(B), dtype=torch.float64, device=dev)
    per_element = 1 if sigma.dim() == 4 else 0          # [B,M,Hy,Wy] (spatial_params) vs [B,M]
    sigma = _f32c(sigma, "rate")
    nu = _f32c(nu, "rate")
    _lib.check(_lib.load().dsic_rate(_p(y), _p(z), _p(y_noisy), _p(z_noisy), _p(sigma), _p(nu),
                                     _p(z_log_sigma), _p(y_hat), _p(y_t), _p(z_t), _p(nll_y), _p(nll_z),
                                     _p(sums), _p(work), B, Hy * Wy, M, Hz * Wz, N, per_element, _stream()),
               "rate")
    return {"y_hat_nhwc": y_hat, "y_tilde": y_t, "z_tilde": z_t, "nll_y": nll_y, "nll_z": nll_z,
            "sums": sums}


def student_t_bits(x, sigma, nu):
    """StudentT.neg_log2_prob on NCHW x; sigma/nu full-shape or spatially constant."""
    x = _f32c(x, "student_t_bits")
    B, C, H, W = x.shape
    per_channel = 0
    if sigma.dim() == 4 and sigma.stride(2) == 0 and sigma.stride(3) == 0 \
            and nu.dim() == 4 and nu.stride(2) == 0 and nu.stride(3) == 0:
        sigma = sigma[:, :, 0, 0].expand(B, C)
        nu = nu[:, :, 0, 0].expand(B, C)
        per_channel = 1
    else:
        sigma = sigma.expand_as(x)
        nu = nu.expand_as(x)
    sigma = _f32c(sigma, "student_t_bits")
    nu = _f32c(nu, "student_t_bits")
    out = torch.empty_like(x)
    _lib.check(_lib.load().dsic_student_t_bits(_p(x), _p(sigma), _p(nu), _p(out), x.numel(), H * W,
                                               per_channel, _stream()), "student_t_bits")
    return out


def gaussian_bits(x, log_sigma):
    x = _f32c(x, "gaussian_bits")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().dsic_gaussian_bits(_p(x), _p(_f32c(log_sigma, "gaussian_bits")), _p(out), B, C,
                                              H * W, _stream()), "gaussian_bits")
    return out


def sigma_nu_spatial(log_sigma_nhwc, log_nu_nhwc, min_nu, max_nu):
    """spatial_params head outputs (NHWC) -> sigma, nu NCHW [B,M,H,W] (model.py:49-51)."""
    ls = _f32c(log_sigma_nhwc, "sigma_nu_spatial")
    ln = _f32c(log_nu_nhwc, "sigma_nu_spatial")
    B, H, W, M = ls.shape
    sigma = torch.empty((B, M, H, W), dtype=torch.float32, device=ls.device)
    nu = torch.empty_like(sigma)
    _lib.check(_lib.load().dsic_sigma_nu_spatial(_p(ls), _p(ln), _p(sigma), _p(nu), B, H * W, M, float(min_nu),
                                                 float(max_nu), _stream()), "sigma_nu_spatial")
    return sigma, nu
