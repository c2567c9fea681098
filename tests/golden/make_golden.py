#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box has none):

    python tests/golden/make_golden.py

The reference's code/modelv2/model.py is imported as-is (read-only tree, no
bytecode written).  `piq` is not installed and is only used inside the
dist="msssim" branch of rate_distortion_loss (model.py:97), so an empty stub
module satisfies the import at model.py:6.  The trained checkpoints are absent
(.MISSING_LARGE_BLOBS), so the reference is loaded with the deterministic
synthetic state_dict of dsic_amd.synthetic (strict load: the key set and every
shape must match the reference's own).

Fixtures are data only: inputs are regenerated from the hash generator, the
files hold the reference's outputs (sums, per-channel parameters, integer
latents, sampled activations).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/code/modelv2")
sys.modules.setdefault("piq", types.ModuleType("piq"))

import layers as ref_layers            # noqa: E402
import distributions as ref_dist       # noqa: E402
from model import CompressionModel, rate_distortion_loss  # noqa: E402

from dsic_amd import synthetic as S    # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

# (name, batch, channels, H, W, weight_seed, first_patch_index)
CASES = [
    ("b1_64x64_s1", 1, 3, 64, 64, 1, 0),
    ("b2_128x96_s1", 2, 3, 128, 96, 1, 10),
    ("b2_128x96_s2", 2, 3, 128, 96, 2, 20),
    ("b1_256x256_s1", 1, 3, 256, 256, 1, 0),
    ("b1_256x256_s2", 1, 3, 256, 256, 2, 1),
    ("b1_4ch_128x128_s1", 1, 4, 128, 128, 1, 30),
    ("b3_48x80_s3", 3, 3, 48, 80, 3, 40),
]
# spatial_params=True (layers.py:127-129, model.py:49-51): per-element sigma/nu heads.  The
# reference's own branch needs H and W to be multiples of 64 (h_s output = 4x the z grid
# must equal the y grid; other sizes raise a shape error inside distributions.py:28).
SPATIAL_CASES = [
    ("spatial_b2_128x64_s1", 2, 3, 128, 64, 1, 50),
    ("spatial_b1_192x192_s2", 1, 3, 192, 192, 2, 60),
]

N_SAMPLES = 64


def sample_idx(numel, tag):
    u = S.hash_uniform(N_SAMPLES, 777, S._stream_id(tag))
    return np.minimum((u.astype(np.float64) * numel).astype(np.int64), numel - 1)


def build_reference(in_ch, seed, spatial=False):
    m = CompressionModel(N=128, M=192, spatial_params=spatial, min_nu=2,
                         max_nu=100.0)
    if in_ch != 3:
        # config 5: same architecture, first/last layer re-sized
        m.g_a.g_a[0] = ref_layers.conv(in_ch, 128, 3, 1)
        m.g_s.g_s[12] = torch.nn.ConvTranspose2d(128, in_ch, 5, 2, 2,
                                                 output_padding=1)
    sd = S.make_state_dict(seed=seed, in_ch=in_ch, spatial_params=spatial)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()},
                      strict=True)
    return m.eval()


def run_case(name, B, C, H, W, seed, first, spatial=False):
    m = build_reference(C, seed, spatial)
    x = torch.from_numpy(S.make_patches(first, B, H, W, C))
    acts = {}
    hooks = []

    def tap(tag):
        def fn(_mod, _inp, out):
            acts[tag] = out.detach().clone()
        return fn

    # layer outputs after each fused (conv, activation) pair
    for i in range(8):
        mod = m.g_a.g_a[2 * i + 1] if i < 7 else m.g_a.g_a[14]
        hooks.append(mod.register_forward_hook(tap(f"g_a.{2 * i}")))
    for i in range(7):
        mod = m.g_s.g_s[2 * i + 1] if i < 6 else m.g_s.g_s[12]
        hooks.append(mod.register_forward_hook(tap(f"g_s.{2 * i}")))
    for idx in (0, 2, 4):
        hooks.append(m.h_a.h_a[idx + 1].register_forward_hook(tap(f"h_a.{idx}")))
    hooks.append(m.h_a.h_a[6].register_forward_hook(tap("h_a.6")))
    hooks.append(m.h_s.h_s[1].register_forward_hook(tap("h_s.0")))
    hooks.append(m.h_s.h_s[3].register_forward_hook(tap("h_s.2")))

    with torch.no_grad():
        out = m(x, quant_mode="round")
        _, R, D = rate_distortion_loss(out, x, lambda_rd=1.0, dist="mse")
    for h in hooks:
        h.remove()

    rec = {
        "meta": np.array([B, C, H, W, seed, first], dtype=np.int64),
        "sum_nll_y": out["nll_y"].double().sum(dim=(1, 2, 3)).numpy(),
        "sum_nll_z": out["nll_z"].double().sum(dim=(1, 2, 3)).numpy(),
        "R_clamped": np.array([R.item()], dtype=np.float64),
        "mse": np.array([D.item()], dtype=np.float64),
        "sigma": (out["sigma"].numpy().copy() if spatial else out["sigma"][:, :, 0, 0].numpy().copy()),
        "nu": (out["nu"].numpy().copy() if spatial else out["nu"][:, :, 0, 0].numpy().copy()),
        "spatial": np.array([int(spatial)], dtype=np.int64),
        "y_tilde": out["y_tilde"].numpy().astype(np.int16),
        "z_tilde": out["z_tilde"].numpy().astype(np.int16),
        "x_hat_crop": out["x_hat"][:, :, :32, :32].numpy().copy(),
        "x_hat_mean": out["x_hat"].double().mean(dim=(1, 2, 3)).numpy(),
    }
    assert np.array_equal(rec["y_tilde"].astype(np.float32), out["y_tilde"].numpy())
    acts["y"] = out["y"]
    acts["z"] = out["z"]
    acts["x_hat"] = out["x_hat"]
    acts["nll_y"] = out["nll_y"]
    acts["nll_z"] = out["nll_z"]
    for tag, a in acts.items():
        flat = a.reshape(-1)
        idx = sample_idx(flat.numel(), name + tag)
        rec[f"act/{tag}/shape"] = np.array(a.shape, dtype=np.int64)
        rec[f"act/{tag}/mean"] = np.array([a.double().mean().item()])
        rec[f"act/{tag}/absmean"] = np.array([a.double().abs().mean().item()])
        rec[f"act/{tag}/idx"] = idx
        rec[f"act/{tag}/val"] = flat[torch.from_numpy(idx)].numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"forward_{name}.npz"), **rec)
    bpp = (rec["sum_nll_y"] + rec["sum_nll_z"]) / (H * W)
    print(f"{name}: bpp={bpp}, y in [{rec['y_tilde'].min()},{rec['y_tilde'].max()}]"
          f" z in [{rec['z_tilde'].min()},{rec['z_tilde'].max()}] mse={D.item():.5f}")


def run_units():
    """Single reference ops on tiny tensors with full outputs."""
    rec = {}
    # GDN forward / inverse (layers.py:6-27) with perturbed beta/gamma
    for inverse in (False, True):
        g = ref_layers.GDN(16, inverse=inverse)
        beta = np.sqrt(0.5 + S.hash_uniform(16, 5, 1) + S.REPARAM_OFFSET).astype(np.float32)
        gam = np.sqrt(0.02 + 0.28 * S.hash_uniform(16, 5, 2) + S.REPARAM_OFFSET).astype(np.float32)
        with torch.no_grad():
            g.beta.copy_(torch.from_numpy(beta))
            g.gamma_conv.weight.copy_(torch.from_numpy(gam).view(16, 1, 1, 1))
            x = torch.from_numpy((S.hash_uniform(2 * 16 * 5 * 7, 5, 3) * 8 - 4)
                                 .reshape(2, 16, 5, 7))
            y = g(x)
        tag = "igdn" if inverse else "gdn"
        rec[f"{tag}/beta"] = beta
        rec[f"{tag}/gamma"] = gam
        rec[f"{tag}/x"] = x.numpy()
        rec[f"{tag}/y"] = y.numpy()
    # conv() helper geometries (layers.py:29-31)
    for k, s, ci, co, h, w in ((3, 1, 8, 16, 9, 11), (5, 2, 8, 16, 9, 11),
                               (5, 2, 16, 8, 12, 6), (1, 1, 8, 8, 1, 1)):
        c = ref_layers.conv(ci, co, k, s)
        wt = (S.hash_uniform(co * ci * k * k, 6, k * 10 + s) - 0.5).reshape(co, ci, k, k)
        bs = S.hash_uniform(co, 6, 100 + k * 10 + s) - 0.5
        x = (S.hash_uniform(2 * ci * h * w, 6, 200 + k * 10 + s) - 0.5).reshape(2, ci, h, w)
        with torch.no_grad():
            c.weight.copy_(torch.from_numpy(wt))
            c.bias.copy_(torch.from_numpy(bs))
            y = c(torch.from_numpy(x))
        tag = f"conv_k{k}s{s}_{ci}to{co}_{h}x{w}"
        rec[tag + "/w"], rec[tag + "/b"], rec[tag + "/x"], rec[tag + "/y"] = wt, bs, x, y.numpy()
    # ConvTranspose2d(ci, co, 5, 2, 2, output_padding=1) as used at layers.py:83
    for ci, co, h, w in ((8, 16, 5, 7), (16, 3, 4, 4)):
        c = torch.nn.ConvTranspose2d(ci, co, 5, 2, 2, output_padding=1)
        wt = (S.hash_uniform(ci * co * 25, 7, ci) - 0.5).reshape(ci, co, 5, 5)
        bs = S.hash_uniform(co, 7, 100 + ci) - 0.5
        x = (S.hash_uniform(2 * ci * h * w, 7, 200 + ci) - 0.5).reshape(2, ci, h, w)
        with torch.no_grad():
            c.weight.copy_(torch.from_numpy(wt))
            c.bias.copy_(torch.from_numpy(bs))
            y = c(torch.from_numpy(x))
        tag = f"convT_{ci}to{co}_{h}x{w}"
        rec[tag + "/w"], rec[tag + "/b"], rec[tag + "/x"], rec[tag + "/y"] = wt, bs, x, y.numpy()
    # priors (distributions.py:20-31, 39-46)
    st = ref_dist.StudentT()
    xs = torch.round(torch.from_numpy(S.hash_uniform(2 * 6 * 4 * 4, 8, 1) * 30 - 15)).reshape(2, 6, 4, 4)
    sig = torch.tensor([1e-4, 0.5, 1.0, 2.5, 40.0, 2e3]).view(1, 6, 1, 1).expand(2, 6, 4, 4)
    nu = torch.tensor([1.1, 2.0, 3.7, 10.0, 100.0, 250.0]).view(1, 6, 1, 1).expand(2, 6, 4, 4)
    with torch.no_grad():
        rec["studentt/x"] = xs.numpy()
        rec["studentt/sigma"] = sig[0, :, 0, 0].numpy().copy()
        rec["studentt/nu"] = nu[0, :, 0, 0].numpy().copy()
        rec["studentt/bits"] = st.neg_log2_prob(xs, sig, nu).numpy()
        fg = ref_dist.FactorizedGaussian(6)
        ls = torch.tensor([-8.0, -1.0, 0.0, 0.7, 2.0, 8.0])
        fg.log_sigma.copy_(ls)
        rec["gauss/log_sigma"] = ls.numpy()
        rec["gauss/bits"] = fg.neg_log2_prob(xs).numpy()
    np.savez_compressed(os.path.join(HERE, "units.npz"), **rec)
    print("units: ", len(rec), "arrays")


if __name__ == "__main__":
    run_units()
    for case in CASES:
        run_case(*case)
    for case in SPATIAL_CASES:
        run_case(*case, spatial=True)
