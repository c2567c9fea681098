#!/usr/bin/env python3
"""CPU emulation of the split-bf16 Winograd contraction, build container only.

Question: may the fp32 MFMA contractions of conv_wino be replaced by bf16 MFMAs on operands split
into bf16 planes (x = hi + mid + lo) with fp32 accumulation, without moving the quantised latents?

The emulation runs the oracle forward (oracle/ref_model.py) with every Winograd-eligible layer
(all 3x3/s1, 5x5/s2 via space-to-depth, ConvTranspose2d(5,2,2,1) via its four 3x3 phases — the
layers conv_wino runs today) replaced by an explicit F(2x2,3x3) Winograd whose 16 position GEMMs
are evaluated
    fp32      : V_p @ U_p in float32                                   (control = today's kernel)
    split6    : V, U split into 3 bf16 planes each (round-to-nearest-even, like v_cvt_pk_bf16_f32),
                products hh, hm, mh, mm, hl, lh accumulated in float32  (the candidate)
    split3    : planes hi, mid only; products hh, hm, mh                (cheaper candidate)
    bf16      : single bf16 plane                                       (what SURVEY §7 rules out)
    f16split3 : two fp16 planes (11 bits each) per operand, the three products of split3, fp16 MFMA at the bf16
                rate; the transformed weights scaled by a power of two per layer (f16split3_noscale: no scale)
and compares y_tilde / z_tilde / bpp with the 9 reference fixtures tests/golden/forward_*.npz.
bf16 x bf16 products are exact in float32, so `a.float() @ b.float()` on bf16-representable
values reproduces an MFMA with fp32 accumulate up to the summation order.

    python tools/split_bf16_emulation.py [mode ...]     -> table on stdout
"""
import glob
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsic_amd import synthetic as S           # noqa: E402
from oracle import ref_model as O             # noqa: E402

torch.set_num_threads(8)

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

MODE = "fp32"


def planes(t, n):
    out, r = [], t
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def planes_f16(t, n, scale=1.0):
    """fp16 planes (11-bit significands, round to nearest even like v_cvt_f16_f32) of t * scale; the planes are
    returned un-scaled again (a power-of-two scale is exact).  Values below 2^-24 * scale flush to zero, subnormal
    planes lose bits: what the scale is for."""
    out, r = [], t * scale
    for _ in range(n):
        p = r.half().float()
        out.append(p / scale)
        r = r - p
    return out


def pos_gemm(V, U):
    """V [16, T, Ci] x U [16, Ci, Co] -> [16, T, Co] under MODE."""
    if MODE in ("f16split3", "f16split3_noscale"):
        # two fp16 planes per operand, the same three products as split3; the transformed weights are scaled by a
        # power of two per layer so that their mid plane (2^-11 of the hi plane) stays a normal fp16 number
        su = 1.0 if MODE.endswith("noscale") else float(2.0 ** np.floor(np.log2(256.0 / float(U.abs().max()))))
        (vh, vm), (uh, um) = planes_f16(V, 2), planes_f16(U, 2, su)
        return (torch.bmm(vh, um) + torch.bmm(vm, uh)) + torch.bmm(vh, uh)
    if MODE == "fp32":
        return torch.bmm(V, U)
    if MODE == "bf16":
        return torch.bmm(planes(V, 1)[0], planes(U, 1)[0])
    if MODE == "split3":
        (vh, vm), (uh, um) = planes(V, 2), planes(U, 2)
        return (torch.bmm(vh, um) + torch.bmm(vm, uh)) + torch.bmm(vh, uh)
    if MODE == "split6":
        (vh, vm, vl), (uh, um, ul) = planes(V, 3), planes(U, 3)
        small = (torch.bmm(vh, ul) + torch.bmm(vl, uh)) + torch.bmm(vm, um)
        return ((small + torch.bmm(vh, um)) + torch.bmm(vm, uh)) + torch.bmm(vh, uh)
    raise ValueError(MODE)


def wino3x3(x, w, bias):
    """3x3 stride-1 'same' correlation, x [B,Ci,H,W], w [Co,Ci,3,3]."""
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    th, tw = He // 2, We // 2
    d = F.unfold(xp, kernel_size=4, stride=2).view(B, Ci, 4, 4, th * tw)          # [B,Ci,4,4,T]
    V = torch.einsum("ij,bcjkt,lk->ilbtc", BT, d, BT).reshape(16, B * th * tw, Ci)  # B^T d B
    U = torch.einsum("ij,ocjk,lk->iloc", G, w, G).reshape(16, Co, Ci).transpose(1, 2).contiguous()
    M = pos_gemm(V.contiguous(), U).view(4, 4, B, th, tw, Co)
    Y = torch.einsum("ij,jkbhwo,lk->bohiwl", AT, M, AT).reshape(B, Co, He, We)
    return Y[:, :, :H, :W] + bias.view(1, -1, 1, 1)


def conv_patched(sd, prefix, x, stride):
    w = O._t(sd, prefix + ".weight")
    b = O._t(sd, prefix + ".bias")
    k = w.shape[-1]
    Co, Ci = w.shape[:2]
    if k == 3 and stride == 1 and Ci % 32 == 0:
        return wino3x3(x, w, b)
    if k == 5 and stride == 2 and Ci % 32 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
        # space-to-depth: X[(a,b,c)][p,q] = x[c][2p+a][2q+b];  g_ab[u][v] = w[2u+a][2v+b] (0 past 4)
        X = torch.cat([x[:, :, a::2, bb::2] for a in (0, 1) for bb in (0, 1)], dim=1)
        w6 = F.pad(w, (0, 1, 0, 1))
        g = torch.cat([w6[:, :, a::2, bb::2] for a in (0, 1) for bb in (0, 1)], dim=1)
        return wino3x3(X, g, b)
    return F.conv2d(x, w, b, stride=stride, padding=(k - 1) // 2)


def convT_patched(sd, prefix, x):
    w = O._t(sd, prefix + ".weight")        # [Ci, Co, 5, 5]
    b = O._t(sd, prefix + ".bias")
    Ci, Co = w.shape[:2]
    if Co % 32 != 0:
        return F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
    B, _, H, W = x.shape
    out = x.new_empty(B, Co, 2 * H, 2 * W)
    w7 = F.pad(w, (0, 2, 0, 2))              # taps 5, 6 = 0
    for py in (0, 1):
        for px in (0, 1):
            rows = [py + 4 - 2 * r for r in range(3)]
            cols = [px + 4 - 2 * c for c in range(3)]
            g = w7[:, :, rows][:, :, :, cols].permute(1, 0, 2, 3).contiguous()   # [Co,Ci,3,3]
            out[:, :, py::2, px::2] = wino3x3(x, g, torch.zeros(Co))
    return out + b.view(1, -1, 1, 1)


def run(mode):
    global MODE
    MODE = mode
    O._conv, O._convT = conv_patched, convT_patched
    rows = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "forward_*.npz"))):
        g = np.load(path)
        B, C, H, W, seed, first = [int(v) for v in g["meta"]]
        spatial = bool(int(g["spatial"][0]))
        sd = S.make_state_dict(seed=seed, in_ch=C, spatial_params=spatial)
        x = torch.from_numpy(S.make_patches(first, B, H, W, C))
        with torch.no_grad():
            out = O.forward(sd, x, "round")
        yf = int((out["y_tilde"].numpy() != g["y_tilde"].astype(np.float32)).sum())
        zf = int((out["z_tilde"].numpy() != g["z_tilde"].astype(np.float32)).sum())
        bpp = (out["nll_y"].double().sum(dim=(1, 2, 3)) + out["nll_z"].double().sum(dim=(1, 2, 3))).numpy() / (H * W)
        ref = (g["sum_nll_y"] + g["sum_nll_z"]) / (H * W)
        dx = float(np.abs(out["x_hat"][:, :, :32, :32].numpy() - g["x_hat_crop"]).max())
        rows.append((os.path.basename(path)[8:-4], B, yf, zf, float(np.abs(bpp - ref).max()), dx))
    return rows


if __name__ == "__main__":
    modes = sys.argv[1:] or ["fp32", "split6", "split3", "bf16"]
    for mode in modes:
        print(f"== {mode}")
        print(f"{'fixture':28s} {'B':>2s} {'y flips':>8s} {'z flips':>8s} {'max |dbpp|':>11s} {'max |dx_hat|':>12s}")
        for name, B, yf, zf, db, dx in run(mode):
            print(f"{name:28s} {B:2d} {yf:8d} {zf:8d} {db:11.2e} {dx:12.2e}", flush=True)
