"""Host-side mirror of the reference's transform modules (code/modelv2/layers.py).

Same class names, constructor arguments, parameter names (so the reference's
state_dict loads with strict=True) and NCHW tensor contract; the compute is the
HIP library.  Every transform also has a `forward_nhwc` used by
CompressionModel.forward to chain layers without leaving NHWC.

Inference only: parameters are plain tensors for the kernels, no autograd.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

import os

from . import ops

# DSIC_WINOGRAD=0 forces the direct implicit-GEMM kernel for every layer (A/B runs)
USE_WINOGRAD = os.environ.get("DSIC_WINOGRAD", "1") != "0"


def wino_bf16() -> bool:
    """The library's arithmetic variant (dsic_split_bf16): True = every contraction on bf16 MFMAs with operands split
    into two bf16 planes (fp32-class results, csrc/conv_wino_bf16.hip; the default), False = fp32-input MFMAs.  One
    switch for the first layer, the Winograd layers and the image layer; DSIC_WINO_BF16=0 only sets its initial value."""
    from . import lib as _lib
    return bool(_lib.load().dsic_split_bf16())


def set_wino_bf16(on: bool) -> None:
    """Switches the variant at run time.  Packed weights are cached per variant (_ConvBase._key), so layers built
    before the switch follow it on their next call."""
    from . import lib as _lib
    _lib.check(_lib.load().dsic_set_split_bf16(1 if on else 0), "set_split_bf16")


def __getattr__(name):   # layers.WINO_BF16: the current variant (read-only view of the switch)
    if name == "WINO_BF16":
        return wino_bf16()
    raise AttributeError(name)


class _GammaConv(nn.Module):
    """Holder for `gamma_conv.weight` [C,1,1,1] (layers.py:15-17)."""

    def __init__(self, channels, init):
        super().__init__()
        self.weight = nn.Parameter(init.view(channels, 1, 1, 1).clone(), requires_grad=False)


class GDN(nn.Module):
    """Diagonal GDN / IGDN (layers.py:6-27): x / sqrt(beta_c + gamma_c x^2)."""

    def __init__(self, channels, inverse=False, beta_min=1e-6, gamma_init=0.1,
                 reparam_offset=2 ** -18):
        super().__init__()
        self.inverse = inverse
        self.reparam_offset = reparam_offset
        self.beta = nn.Parameter(torch.sqrt(torch.ones(channels) + reparam_offset),
                                 requires_grad=False)
        # `gamma` [C,C] is never read by the reference's forward (layers.py:13 vs
        # :19-27); it exists only so that checkpoints load strictly.
        gamma = torch.sqrt(torch.eye(channels) * gamma_init + reparam_offset)
        self.gamma = nn.Parameter(gamma, requires_grad=False)
        self.gamma_conv = _GammaConv(channels, gamma.diag())

    def effective(self):
        """(beta_eff, gamma_eff) = (beta^2 - off, w^2 - off), layers.py:20-21.

        Cached until a parameter is rewritten or moved (load_state_dict, .to())."""
        w = self.gamma_conv.weight
        key = (self.beta._version, self.beta.data_ptr(), w._version, w.data_ptr())
        if getattr(self, "_eff_key", None) != key:
            beta = self.beta ** 2 - self.reparam_offset
            gamma = (w ** 2 - self.reparam_offset).reshape(-1)
            self._eff = (beta.contiguous(), gamma.contiguous())
            self._eff_key = key
        return self._eff

    @torch.no_grad()
    def forward(self, x):
        beta, gamma = self.effective()
        return ops.gdn_nchw(x, beta, gamma, self.inverse)


class _ConvBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._packed = None
        self._packed_key = None

    def _key(self):
        w = self.weight
        return (w._version, w.data_ptr(), str(w.device), wino_bf16())

    def packed(self):
        key = self._key()
        if self._packed is None or self._packed_key != key:
            self._packed = self._pack()
            self._packed_key = key
        return self._packed


class Conv2d(_ConvBase):
    """nn.Conv2d(in,out,k,stride,padding=(k-1)//2) as built by conv() (layers.py:29-31)."""

    def __init__(self, in_ch, out_ch, k, stride=1):
        super().__init__()
        self.in_channels, self.out_channels = in_ch, out_ch
        self.kernel_size, self.stride = k, stride
        # layers on few tiles per image may share a tile's input channels between workgroups (ops.WINO_SPLITK);
        # HyperAnalysis switches it off: its launches run on the side stream, where every extra launch waits for
        # compute units at a kernel boundary of the main stream
        self.split_k = True
        w = torch.empty(out_ch, in_ch, k, k)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_ch * k * k)
        self.weight = nn.Parameter(w, requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_ch).uniform_(-bound, bound), requires_grad=False)

    def _pack(self):
        if self.kernel_size == 1:
            # 1x1 heads run inside dsic_hyper_params: input-major [Cin][Cout]
            return self.weight.view(self.out_channels, self.in_channels).t().contiguous()
        return ops.pack_conv_weight(self.weight)

    @property
    def use_winograd(self):
        """3x3 stride-1 layers with MFMA-friendly channel counts run as Winograd F(2x2,3x3)."""
        return (USE_WINOGRAD and self.kernel_size == 3 and self.stride == 1 and self.in_channels % 32 == 0
                and self.out_channels % 4 == 0 and 64 <= self.out_channels <= 128)

    @property
    def use_winograd_s2(self):
        """5x5 stride-2 layers run as a 3x3 Winograd conv over the space-to-depth input (4*Cin
        channels) when the producing layer can write that layout."""
        # both Winograd kernels need 4*Cin as a multiple of 128 over the space-to-depth input (Cin % 32 == 0); other
        # widths (cfg.MODEL.N is configurable in the reference's train.py) take the direct implicit GEMM
        if not (USE_WINOGRAD and self.kernel_size == 5 and self.stride == 2 and self.in_channels % 32 == 0
                and self.out_channels % 4 == 0 and self.out_channels >= 64):
            return False
        # wider outputs (g_a.14: 128 -> 192, layers.py:72) run as channel slices of <= 128 into one tensor;
        # only the split-bf16 kernel can store a slice
        return self.out_channels <= 128 or (wino_bf16() and self.out_channels <= 256 and 4 * self.in_channels >= 64)

    def _cout_slices(self):
        """(lo, hi) output-channel slices of at most 128 (multiples of 32 except the last)."""
        n = self.out_channels
        return [(lo, min(lo + 128, n)) for lo in range(0, n, 128)]

    def packed_wino_slices(self):
        """per Cout slice: (lo, hi, bf16 planes, bias, ...) for out_channels > 128 (space-to-depth 5x5/s2 only)"""
        key = self._key()
        if getattr(self, "_wino_sl", None) is None or self._wino_sl_key != key:
            sl = []
            for lo, hi in self._cout_slices():
                u = ops.pack_wino_s2_weight(self.weight[lo:hi].contiguous())
                u = ops.split_wino_weight_bf16(u, hi - lo, 4 * self.in_channels, 1)
                sl.append((lo, hi, u, self.bias[lo:hi].contiguous()))
            self._wino_sl, self._wino_sl_key = sl, key
        return self._wino_sl

    def packed_wino(self):
        key = self._key()
        if getattr(self, "_wino", None) is None or self._wino_key != key:
            self._wino = (ops.pack_wino_s2_weight(self.weight) if self.kernel_size == 5
                          else ops.pack_wino_weight(self.weight))
            cin = self.in_channels * (4 if self.kernel_size == 5 else 1)
            if wino_bf16() and cin >= 64:
                self._wino = ops.split_wino_weight_bf16(self._wino, self.out_channels, cin, 1)
            self._wino_key = key
        return self._wino

    def run_nhwc(self, x, act=ops.ACT_NONE, gdn=None, x_is_s2d=False, s2d_out=False):
        """x_is_s2d: x is the space-to-depth image of this layer's input; s2d_out: write the
        output space-to-depth (only the Winograd paths can)."""
        beta = gamma = None
        if gdn is not None:
            beta, gamma = gdn.effective()
        if x_is_s2d and self.out_channels > 128:
            B, H2, W2, _ = x.shape
            assert not s2d_out
            out = torch.empty((B, H2, W2, self.out_channels), dtype=torch.float32, device=x.device)
            for lo, hi, u, bias in self.packed_wino_slices():
                ops.conv3x3_wino_nhwc(x, u, bias, hi - lo, act, None if beta is None else beta[lo:hi].contiguous(),
                                      None if gamma is None else gamma[lo:hi].contiguous(), out=out, s2d_in=True,
                                      out_coff=lo, split_k=self.split_k,
                                      algo_flops=2.0 * B * H2 * W2 * (hi - lo) * self.in_channels * 25)
            return out
        if x_is_s2d:
            B, H2, W2, _ = x.shape
            return ops.conv3x3_wino_nhwc(x, self.packed_wino(), self.bias, self.out_channels, act, beta, gamma,
                                         s2d_out=s2d_out, s2d_in=True, split_k=self.split_k,
                                         algo_flops=2.0 * B * H2 * W2 * self.out_channels * self.in_channels * 25)
        if self.use_winograd and x.shape[-1] == self.in_channels:
            return ops.conv3x3_wino_nhwc(x, self.packed_wino(), self.bias, self.out_channels, act, beta, gamma,
                                         s2d_out=s2d_out, split_k=self.split_k)
        assert not s2d_out
        return ops.conv2d_nhwc(x, self.packed(), self.bias, self.out_channels, self.kernel_size,
                               self.stride, act, beta, gamma, cin_real=self.in_channels)

    def can_write_s2d(self, x_is_s2d):
        return x_is_s2d or self.use_winograd

    @torch.no_grad()
    def forward(self, x):
        return ops.nhwc_to_nchw(self.run_nhwc(_to_nhwc(x)))


def conv(in_ch, out_ch, k, stride=1):
    """layers.py:29-31."""
    return Conv2d(in_ch, out_ch, k, stride)


class ConvTranspose2d(_ConvBase):
    """nn.ConvTranspose2d(in,out,5,2,2,output_padding=1) (layers.py:83)."""

    def __init__(self, in_ch, out_ch, kernel_size=5, stride=2, padding=2, output_padding=1):
        super().__init__()
        if (kernel_size, stride, padding, output_padding) != (5, 2, 2, 1):
            raise ValueError("only ConvTranspose2d(k=5, s=2, p=2, output_padding=1) is on the hot path")
        self.in_channels, self.out_channels = in_ch, out_ch
        w = torch.empty(in_ch, out_ch, 5, 5)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(out_ch * 25)
        self.weight = nn.Parameter(w, requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_ch).uniform_(-bound, bound), requires_grad=False)

    @property
    def to_image(self):
        return self.out_channels % 8 != 0

    @property
    def use_winograd(self):
        return (USE_WINOGRAD and not self.to_image and self.in_channels % 32 == 0
                and self.out_channels % 4 == 0 and 64 <= self.out_channels <= 128)

    def _pack(self):
        if self.to_image:
            return ops.pack_convT_image_weight(self.weight)
        if self.use_winograd:
            u = ops.pack_wino_convT_weight(self.weight)
            if wino_bf16() and self.in_channels >= 64:
                u = ops.split_wino_weight_bf16(u, self.out_channels, self.in_channels, 4)
            return u
        return ops.pack_convT_weight(self.weight)

    def run_nhwc(self, x, act=ops.ACT_NONE, gdn=None):
        if self.to_image:   # returns NCHW image
            return ops.conv_transpose2d_image(x, self.packed(), self.bias, self.out_channels)
        beta = gamma = None
        if gdn is not None:
            beta, gamma = gdn.effective()
        if self.use_winograd:
            return ops.conv_transpose2d_wino_nhwc(x, self.packed(), self.bias, self.out_channels, act, beta, gamma)
        return ops.conv_transpose2d_nhwc(x, self.packed(), self.bias, self.out_channels, act, beta, gamma)

    @torch.no_grad()
    def forward(self, x):
        y = self.run_nhwc(_to_nhwc(x))
        return y if self.to_image else ops.nhwc_to_nchw(y)


def _f32_image(x):
    if x.dim() != 4:
        raise ValueError(f"expected [N,C,H,W], got {tuple(x.shape)}")
    return x


def _to_nhwc(x):
    """NCHW -> NHWC with the channel count padded to a multiple of 8."""
    if x.dim() != 4:
        raise ValueError(f"expected [N,C,H,W], got {tuple(x.shape)}")
    C = x.shape[1]
    if C <= 8 and C % 8 != 0:
        return ops.image_to_nhwc8(x)
    if C % 8 != 0:
        raise ValueError(f"channel count {C} must be <= 8 or a multiple of 8")
    return ops.nchw_to_nhwc(x)


class _Chain(nn.Sequential):
    """nn.Sequential whose (conv, GDN|ReLU) pairs run as one fused kernel."""

    @staticmethod
    def _wants_s2d(mods, j, H, W):
        """Does the conv that consumes the output of the layer ending before index j take
        space-to-depth input?  (H, W: spatial size of that output.)"""
        nxt = mods[j] if j < len(mods) else None
        return isinstance(nxt, Conv2d) and nxt.use_winograd_s2 and H % 2 == 0 and W % 2 == 0

    def forward_from_image(self, x_nchw, taps=None):
        """Like forward_nhwc but from an NCHW image: a leading conv(3|4 -> <=128, 3, 1)
        runs as the dedicated first-layer kernel (K = 9*Cimg, no channel padding)."""
        mods = list(self)
        m = mods[0] if mods else None
        u8 = x_nchw.dtype == torch.uint8          # decoded image bytes [B,H,W,C]: to_tensor is fused into the kernel
        cin = x_nchw.shape[3] if u8 else x_nchw.shape[1]
        first_ok = (isinstance(m, Conv2d) and m.kernel_size == 3 and m.stride == 1 and m.in_channels in (3, 4)
                    and m.out_channels <= 128 and m.out_channels % 4 == 0 and cin == m.in_channels)
        if u8 and not first_ok:
            x_nchw, u8 = ops.to_tensor_u8(x_nchw), False
        if first_ok:
            nxt = mods[1] if len(mods) > 1 else None
            H, W = (x_nchw.shape[1], x_nchw.shape[2]) if u8 else (x_nchw.shape[2], x_nchw.shape[3])
            if isinstance(nxt, GDN) and not nxt.inverse:
                beta, gamma = nxt.effective()
                s2d = self._wants_s2d(mods, 2, H, W)
                y = ops.conv_first_nchw(x_nchw, m.weight, m.bias, ops.ACT_GDN, beta, gamma, s2d_out=s2d)
                start = 2
            elif isinstance(nxt, nn.ReLU):
                s2d = self._wants_s2d(mods, 2, H, W)
                y = ops.conv_first_nchw(x_nchw, m.weight, m.bias, ops.ACT_RELU, s2d_out=s2d)
                start = 2
            else:
                s2d = self._wants_s2d(mods, 1, H, W)
                y = ops.conv_first_nchw(x_nchw, m.weight, m.bias, s2d_out=s2d)
                start = 1
            if taps is not None:
                taps.append(ops.depth_to_space(y) if s2d else y)
            return self.forward_nhwc(y, taps, start, x_is_s2d=s2d)
        return self.forward_nhwc(_to_nhwc(x_nchw), taps)

    def forward_nhwc(self, x, taps=None, start=0, x_is_s2d=False):
        mods = list(self)
        i = start
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            out_s2d = False
            if isinstance(m, Conv2d):
                fused = isinstance(nxt, (GDN, nn.ReLU))
                # spatial size of this layer's output
                if x_is_s2d:
                    Ho, Wo = x.shape[1], x.shape[2]
                else:
                    Ho, Wo = -(-x.shape[1] // m.stride), -(-x.shape[2] // m.stride)
                out_s2d = m.can_write_s2d(x_is_s2d) and self._wants_s2d(mods, i + (2 if fused else 1), Ho, Wo)
                if isinstance(nxt, GDN):
                    x = m.run_nhwc(x, ops.ACT_IGDN if nxt.inverse else ops.ACT_GDN, nxt, x_is_s2d, out_s2d)
                elif isinstance(nxt, nn.ReLU):
                    x = m.run_nhwc(x, ops.ACT_RELU, None, x_is_s2d, out_s2d)
                else:
                    x = m.run_nhwc(x, ops.ACT_NONE, None, x_is_s2d, out_s2d)
                i += 2 if fused else 1
            elif isinstance(m, ConvTranspose2d):
                assert not x_is_s2d
                if isinstance(nxt, GDN):
                    x = m.run_nhwc(x, ops.ACT_IGDN if nxt.inverse else ops.ACT_GDN, nxt)
                    i += 2
                elif isinstance(nxt, nn.ReLU):
                    x = m.run_nhwc(x, ops.ACT_RELU)
                    i += 2
                else:
                    x = m.run_nhwc(x)
                    i += 1
            elif isinstance(m, GDN):
                assert not x_is_s2d
                x = ops.nchw_to_nhwc(m(ops.nhwc_to_nchw(x)))
                i += 1
            elif isinstance(m, nn.ReLU):
                x = torch.relu(x)
                i += 1
            else:  # pragma: no cover
                raise TypeError(f"unsupported module {type(m).__name__}")
            x_is_s2d = out_s2d
            if taps is not None:
                taps.append(ops.depth_to_space(x) if x_is_s2d else x)
        assert not x_is_s2d
        return x

    @torch.no_grad()
    def forward(self, x):
        y = self.forward_nhwc(_to_nhwc(x))
        last = list(self)[-1]
        if isinstance(last, ConvTranspose2d) and last.to_image:
            return y
        return ops.nhwc_to_nchw(y)


class AnalysisTransform(nn.Module):
    """layers.py:46-76: 8 convs (3/1,5/2 alternating) + 7 GDN, /16."""

    def __init__(self, N=128, M=192, in_ch=3):
        super().__init__()
        self.g_a = _Chain(
            conv(in_ch, N, 3, 1), GDN(N),
            conv(N, N, 5, 2), GDN(N),
            conv(N, N, 3, 1), GDN(N),
            conv(N, N, 5, 2), GDN(N),
            conv(N, N, 3, 1), GDN(N),
            conv(N, N, 5, 2), GDN(N),
            conv(N, N, 3, 1), GDN(N),
            conv(N, M, 5, 2),
        )

    def forward_nhwc(self, x, taps=None):
        return self.g_a.forward_nhwc(x, taps)

    def forward_from_image(self, x_nchw, taps=None):
        return self.g_a.forward_from_image(x_nchw, taps)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.g_a.forward_from_image(_f32_image(x)))


class SynthesisTransform(nn.Module):
    """layers.py:78-101: 4 convT + 3 conv + 6 IGDN, x16."""

    def __init__(self, N=128, M=192, out_ch=3):
        super().__init__()
        self.g_s = _Chain(
            ConvTranspose2d(M, N, 5, 2, 2, output_padding=1), GDN(N, inverse=True),
            conv(N, N, 3, 1), GDN(N, inverse=True),
            ConvTranspose2d(N, N, 5, 2, 2, output_padding=1), GDN(N, inverse=True),
            conv(N, N, 3, 1), GDN(N, inverse=True),
            ConvTranspose2d(N, N, 5, 2, 2, output_padding=1), GDN(N, inverse=True),
            conv(N, N, 3, 1), GDN(N, inverse=True),
            ConvTranspose2d(N, out_ch, 5, 2, 2, output_padding=1),
        )

    def forward_nhwc(self, y_hat, taps=None):
        """NHWC latents -> NCHW image."""
        return self.g_s.forward_nhwc(y_hat, taps)

    def forward(self, y_hat):
        return self.g_s(y_hat)


class HyperAnalysis(nn.Module):
    """layers.py:104-116."""

    def __init__(self, M=192, N=128):
        super().__init__()
        self.h_a = _Chain(
            conv(M, N, 3, 1), nn.ReLU(inplace=True),
            conv(N, N, 3, 1), nn.ReLU(inplace=True),
            conv(N, N, 5, 2), nn.ReLU(inplace=True),
            conv(N, N, 5, 2),
        )
        for m in self.h_a:
            if isinstance(m, Conv2d):
                m.split_k = False

    def forward_nhwc(self, y, taps=None):
        return self.h_a.forward_nhwc(y, taps)

    def forward(self, y):
        return self.h_a(y)


class HyperSynthesis(nn.Module):
    """layers.py:118-152, non-spatial heads (the only branch any reference script uses)."""

    def __init__(self, N=128, M=128, spatial_params=False):
        super().__init__()
        self.spatial_params = spatial_params
        self.N, self.M = N, M
        self.h_s = _Chain(
            ConvTranspose2d(N, N, 5, 2, 2, output_padding=1), nn.ReLU(inplace=True),
            ConvTranspose2d(N, N, 5, 2, 2, output_padding=1), nn.ReLU(inplace=True),
        )
        if spatial_params:      # layers.py:127-129: per-element heads
            self.to_sigma = conv(N, M, 3, 1)
            self.to_nu = conv(N, M, 3, 1)
        else:                   # layers.py:130-139: global per-channel heads
            self.pool = nn.AdaptiveAvgPool2d(1)
            self.mlp_sigma = nn.Sequential(Conv2d(N, N, 1), nn.ReLU(), Conv2d(N, M, 1))
            self.mlp_nu = nn.Sequential(Conv2d(N, N, 1), nn.ReLU(), Conv2d(N, M, 1))

    def params_nhwc(self, z_hat_nhwc, min_nu, max_nu, taps=None):
        """spatial_params=False: -> ((log_sigma, log_nu, sigma, nu) each [B,M], (Ht, Wt)).
        spatial_params=True:  -> ((log_sigma, log_nu) NHWC [B,Ht,Wt,M], sigma, nu NCHW), (Ht, Wt)."""
        t = self.h_s.forward_nhwc(z_hat_nhwc, taps)
        if self.spatial_params:
            ls = self.to_sigma.run_nhwc(t)
            ln = self.to_nu.run_nhwc(t)
            sigma, nu = ops.sigma_nu_spatial(ls, ln, min_nu, max_nu)
            return (ls, ln, sigma, nu), (t.shape[1], t.shape[2])
        s0, s2 = self.mlp_sigma[0], self.mlp_sigma[2]
        n0, n2 = self.mlp_nu[0], self.mlp_nu[2]
        outs = ops.hyper_params(t, s0.packed(), s0.bias, s2.packed(), s2.bias, n0.packed(), n0.bias,
                                n2.packed(), n2.bias, self.M, min_nu, max_nu)
        return outs, (t.shape[1], t.shape[2])

    @torch.no_grad()
    def forward(self, z):
        # clamp bounds are irrelevant for the log outputs returned here
        (log_sigma, log_nu, _, _), (Ht, Wt) = self.params_nhwc(_to_nhwc(z), 0.0, float("inf"))
        if self.spatial_params:
            return ops.nhwc_to_nchw(log_sigma), ops.nhwc_to_nchw(log_nu)
        B = z.shape[0]
        return (log_sigma.view(B, self.M, 1, 1).expand(-1, -1, Ht, Wt),
                log_nu.view(B, self.M, 1, 1).expand(-1, -1, Ht, Wt))
