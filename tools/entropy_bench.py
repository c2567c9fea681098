#!/usr/bin/env python3
"""Timing of the entropy path stages at the bench shape (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsic_amd import entropy, synthetic as S
from dsic_amd.model import CompressionModel

B = int(os.environ.get("B", "64"))
sd = S.make_state_dict(seed=1)
m = CompressionModel(min_nu=2).cuda().eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
x = torch.from_numpy(S.make_patches(0, B, 256, 256)).cuda()
out = m(x, "round")
sy, ny = out["sigma"][:, :, 0, 0].contiguous(), out["nu"][:, :, 0, 0].contiguous()
sz = torch.exp(m.z_prior.log_sigma)

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

meta = entropy.latent_support(out["y_tilde"], out["z_tilde"])
print("support   %.3f ms" % timeit(lambda: entropy.latent_support(out["y_tilde"], out["z_tilde"])))
print("tables    %.3f ms" % timeit(lambda: entropy.cdf_tables(sy, ny, sz, meta)))
print("compress  %.3f ms (support+tables+encode)" % timeit(lambda: entropy.compress_latents(out["y_tilde"], out["z_tilde"], sy, ny, sz)))
print("forward   %.3f ms" % timeit(lambda: m(x, "round")))
c = entropy.compress_latents(out["y_tilde"], out["z_tilde"], sy, ny, sz)
print("err", int(c["err"].item()), "bytes/img", float(c["lengths"].sum().item()) / B, "meta0", c["meta"][0].tolist())
