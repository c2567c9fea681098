#!/usr/bin/env python3
"""Timing of custom_decompress and of its range-decode launches at the bench shape (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsic_amd import entropy, ops, lib as _lib, synthetic as S
from dsic_amd.entropy import _p, _stream, _upload_strings, sigma_z_of, DEFAULT_LMAX
from dsic_amd.model import CompressionModel
B = int(os.environ.get("B", "64"))
m = CompressionModel(min_nu=2).cuda().eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.make_state_dict(seed=1).items()})
x = torch.from_numpy(S.make_patches(0, B, 256, 256)).cuda()
c = entropy.custom_compress(m, x)
ref = entropy.custom_decompress(m, c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): entropy.custom_decompress(m, c)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"custom_decompress (incl. H2D): {dt*1e3:.1f} ms per batch of {B} -> {B/dt:.0f} patches/s")

# the two range-decode launches alone
dev = x.device
L = _lib.load()
_, M, Hy, Wy = c["shape_y"]; _, N, Hz, Wz = c["shape_z"]
meta_np = np.array([[c["min_y"][b], c["max_y"][b] - c["min_y"][b] + 1, c["min_z"][b], c["max_z"][b] - c["min_z"][b] + 1]
                    for b in range(B)], dtype=np.int32)
Lmax = max(DEFAULT_LMAX, int(meta_np[:, [1, 3]].max()))
print("support widths L_y min/median/max", int(meta_np[:, 1].min()), int(np.median(meta_np[:, 1])), int(meta_np[:, 1].max()),
      " L_z", int(meta_np[:, 3].min()), int(meta_np[:, 3].max()), " Lmax", Lmax)
meta = torch.from_numpy(meta_np).to(dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
tab_z = torch.zeros((B, N, Lmax), dtype=torch.uint16, device=dev)
_lib.check(L.dsic_cdf_tables_gauss(_p(sigma_z_of(m)), _p(meta), _p(tab_z), B, N, Lmax, _p(err), _stream()), "tables")
zbuf, zlen, zstride = _upload_strings(c["strings"], 0, dev)
z_hat = torch.empty((B, N, Hz, Wz), dtype=torch.float32, device=dev)
dec_z = lambda: _lib.check(L.dsic_range_decode(_p(zbuf), zstride, _p(zlen), 1, 0, _p(meta), 2, _p(tab_z), Lmax, B, N, Hz * Wz, 0,
                                               _p(z_hat), _p(err), _stream()), "decode z")
dec_z()
(_, _, sigma_y, nu_y), _ = m.h_s.params_nhwc(ops.nchw_to_nhwc(z_hat), m.min_nu, m.max_nu)
tab_y = torch.zeros((B, M, Lmax), dtype=torch.uint16, device=dev)
_lib.check(L.dsic_cdf_tables_student(_p(sigma_y.contiguous()), _p(nu_y.contiguous()), _p(meta), _p(tab_y), B, M, Lmax, _p(err),
                                     _stream()), "tables y")
ybuf, ylen, ystride = _upload_strings(c["strings"], 1, dev)
y_hat = torch.empty((B, M, Hy, Wy), dtype=torch.float32, device=dev)
dec_y = lambda: _lib.check(L.dsic_range_decode(_p(ybuf), ystride, _p(ylen), 1, 0, _p(meta), 0, _p(tab_y), Lmax, B, M, Hy * Wy, 0,
                                               _p(y_hat), _p(err), _stream()), "decode y")
for name, f, nsym in (("range_decode z", dec_z, N * Hz * Wz), ("range_decode y", dec_y, M * Hy * Wy)):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name}: {ms:.3f} ms per launch, {ms * 1e6 / nsym:.0f} ns per symbol of a string ({B} strings side by side)")
out = m(x, "round")
print("decoded y equals the encoder's latents:", bool(torch.equal(y_hat, out["y_tilde"])), " err", int(err.item()))
