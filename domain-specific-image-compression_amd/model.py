"""Mirror of code/modelv2/model.py: CompressionModel and rate_distortion_loss.

forward() keeps activations NHWC on the device from the first conv to the last
and launches only HIP kernels of libdsic_hip.so; the returned dict has the
reference's nine keys with the reference's NCHW shapes (model.py:65-72).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .distributions import FactorizedGaussian, StudentT
from .layers import AnalysisTransform, HyperAnalysis, HyperSynthesis, SynthesisTransform, _to_nhwc


class ForwardOutput(dict):
    """The reference's output dict plus device-side per-image sums.

    `sums` is [B,2] float64 = (sum nll_y, sum nll_z) per image, so evaluators can
    form bpp without re-reducing nll tensors on the host (modelseval.py:90-94).
    """
    sums = None
    layer_taps = None


# The hyperprior branch (h_a, round z, h_s, the rate terms and the `after_rate` hook: 9 launches on 4x4..16x16
# latents, a few dozen workgroups each, ~0.5 ms of dependent launches at 256x256) needs only y; synthesis needs only
# round(y).  forward() therefore forks: the branch runs on a second, high-priority HIP stream beside g_s and is
# joined before forward() returns.  DSIC_HYPER_STREAM=0 keeps everything on the caller's stream.
HYPER_STREAM = os.environ.get("DSIC_HYPER_STREAM", "1") != "0"


class CompressionModel(nn.Module):
    def __init__(self, N=128, M=192, spatial_params=False, min_nu=1.1, max_nu=100.0, in_ch=3):
        super().__init__()
        if N % 16 or M % 8 or N < 16 or M < 8:
            raise ValueError(f"CompressionModel: N={N} must be a multiple of 16 and M={M} a multiple of 8 (NHWC tiles of the "
                             "HIP kernels; the reference's widths 128 / 192 and every multiple of 32 run on the Winograd path)")
        self.g_a = AnalysisTransform(N, M, in_ch=in_ch)
        self.g_s = SynthesisTransform(N, M, out_ch=in_ch)
        self.h_a = HyperAnalysis(M, N)
        self.h_s = HyperSynthesis(N, M, spatial_params=spatial_params)
        self.studentT = StudentT()
        self.z_prior = FactorizedGaussian(N)
        self.min_nu = min_nu
        self.max_nu = max_nu
        self.spatial_params = spatial_params
        self.N, self.M = N, M
        self._fork = {}              # (device index, caller's stream) -> (side stream, fork event, join event)
        self._stage_events = None    # pre-created join events (reserve_stage_events)

    def _hyper_fork(self, device, main):
        # one side stream per stream the model is called on: two batches in flight on two streams must not share
        # the fork / join events
        key = (device.index, main.cuda_stream)
        f = self._fork.get(key)
        if f is None:
            with torch.cuda.device(device):
                f = (torch.cuda.Stream(priority=-1), torch.cuda.Event(), torch.cuda.Event())
            self._fork[key] = f
        return f

    @staticmethod
    def quantize(x, mode):
        """model.py:27-35."""
        if mode == "noise":
            return x + torch.empty_like(x).uniform_(-0.5, 0.5)
        elif mode == "round":
            return ops.round_half_even(x)
        else:
            raise ValueError(f"Unknown quant mode: {mode}")

    @torch.no_grad()
    def forward(self, x, quant_mode="noise", collect_taps=False, after_rate=None):
        """model.py:37-72.  x: [B,C,H,W] float32 on the GPU, H and W multiples of 16 — or the decoded
        image bytes, uint8 [B,H,W,C] (PIL / numpy layout): torchvision's to_tensor
        (modelseval.py:66-67) is then fused into the first layer's kernel.

        `after_rate(partial)` (optional) is called once the latents, sigma/nu and
        the rate terms are enqueued and BEFORE synthesis is launched, so a caller
        can start the range coder on a second HIP stream beside g_s."""
        return self.decode_stage(self.encode_stage(x, quant_mode, collect_taps, after_rate, _staged=False))

    @torch.no_grad()
    def encode_stage(self, x, quant_mode="noise", collect_taps=False, after_rate=None, _staged=True):
        """The first half of forward(): g_a, and - forked onto the side stream - h_a, round, h_s, the rate terms and the
        `after_rate` hook (model.py:39-59).  Returns the state decode_stage() turns into forward()'s dict.  A caller that
        streams batches may run encode_stage(i + 1) before decode_stage(i): the range coder of a batch then has the
        synthesis of two batches to hide behind (bench.py)."""
        if quant_mode not in ("noise", "round"):
            raise ValueError(f"Unknown quant mode: {quant_mode}")
        if x.dim() != 4:
            raise ValueError(f"expected [N,C,H,W], got {tuple(x.shape)}")
        B = x.shape[0]
        taps = [] if collect_taps else None
        y = self.g_a.forward_from_image(x, taps)           # [B,H/16,W/16,M]
        y_noisy = self.quantize(y, "noise") if quant_mode == "noise" else None
        # model.py:62: eval mode synthesises from round(y), training from y_tilde
        y_hat = y_noisy if (self.training and y_noisy is not None) else ops.round_half_even(y)

        def hyper_branch():
            z = self.h_a.forward_nhwc(y, taps)             # [B,.,.,N]
            z_noisy = self.quantize(z, "noise") if quant_mode == "noise" else None
            z_in = z_noisy if z_noisy is not None else ops.round_half_even(z)
            (log_sigma, log_nu, sigma, nu), _ = self.h_s.params_nhwc(z_in, self.min_nu, self.max_nu, taps)
            r = ops.rate(y, z, sigma, nu, self.z_prior.log_sigma, y_noisy, z_noisy)
            if after_rate is not None:
                after_rate({"y_tilde": r["y_tilde"], "z_tilde": r["z_tilde"], "sigma": sigma, "nu": nu,
                            "sums": r["sums"]})
            return z, sigma, nu, r

        joined = None
        if HYPER_STREAM:
            main = torch.cuda.current_stream(y.device)
            side, forked, joined = self._hyper_fork(y.device, main)
            if _staged:
                # an event of its own per call: two batches may be between their stages at the same time
                joined = self._stage_events.pop() if self._stage_events else torch.cuda.Event()
            forked.record(main)
            with torch.cuda.stream(side):
                side.wait_event(forked)
                z, sigma, nu, r = hyper_branch()
                joined.record(side)
        else:
            z, sigma, nu, r = hyper_branch()
        # y_noisy rides along: the side stream reads it, so it must outlive the join in decode_stage (it was allocated on
        # the main stream, whose pool would hand the block to g_s while the rate kernel still reads it)
        return {"B": B, "y": y, "y_hat": y_hat, "z": z, "sigma": sigma, "nu": nu, "r": r, "taps": taps, "joined": joined,
                "y_noisy": y_noisy}

    def reserve_stage_events(self, n):
        """create n join events now (bench.py: creating a HIP event inside a timed loop stalls the enqueue thread)"""
        self._stage_events = [torch.cuda.Event() for _ in range(n)]
        for e in self._stage_events:
            e.record()

    @torch.no_grad()
    def decode_stage(self, st):
        """The second half of forward(): g_s on round(y) (model.py:62-63), joined with the hyperprior branch."""
        y, z, sigma, nu, r, taps, B = st["y"], st["z"], st["sigma"], st["nu"], st["r"], st["taps"], st["B"]
        x_hat = self.g_s.forward_nhwc(st["y_hat"], taps)
        if st["joined"] is not None:
            # the branch's tensors live in the side stream's pool; every later use on `main` is ordered after
            # this join, and the next fork waits for `main`, so the pool never recycles them under a reader
            torch.cuda.current_stream(y.device).wait_event(st["joined"])
        Hy, Wy = y.shape[1], y.shape[2]
        out = ForwardOutput({
            "x_hat": x_hat,
            "nll_y": r["nll_y"],
            "nll_z": r["nll_z"],
            "y": ops.nhwc_to_nchw(y), "y_tilde": r["y_tilde"],
            "z": ops.nhwc_to_nchw(z), "z_tilde": r["z_tilde"],
            "sigma": sigma if self.spatial_params else sigma.view(B, self.M, 1, 1).expand(-1, -1, Hy, Wy),
            "nu": nu if self.spatial_params else nu.view(B, self.M, 1, 1).expand(-1, -1, Hy, Wy),
        })
        out.sums = r["sums"]
        out.layer_taps = taps
        return out


def rate_distortion_loss(out, x, lambda_rd=10000.0, dist="mssim"):
    """model.py:75-107: returns (loss, R, D) with R = clamp(sum nll / (N*H*W), 0)."""
    from . import metrics
    N, C, H, W = x.shape
    if dist not in ("mse", "msssim"):
        raise ValueError("dist must be 'mse' or 'msssim'")
    sums = getattr(out, "sums", None)
    if sums is not None:
        total = sums.sum()
    else:
        total = out["nll_y"].double().sum() + out["nll_z"].double().sum()
    R = torch.clamp(total / (N * H * W), min=0.0).float()
    x_hat = out["x_hat"]
    if dist == "mse":
        D = metrics.mse(x_hat, x)
    else:
        if x_hat.shape[2:] != x.shape[2:]:
            # model.py:95-96: the reference resizes x_hat bilinearly when the sizes differ (never the case
            # for inputs padded to a multiple of 16); rare path, plain torch on the device tensors
            x_hat = torch.nn.functional.interpolate(x_hat, size=x.shape[2:], mode="bilinear", align_corners=False)
        D = 1.0 - metrics.ms_ssim(x_hat.clamp(0, 1), x, data_range=1.0, weights=(0.3, 0.5, 0.2))
    loss = lambda_rd * D + R
    return loss, R, D
