"""torch.ops.dsic.*: the forward operators registered with the PyTorch dispatcher (dsic_amd/torch_ops.py) give the same
bits as the direct C-ABI calls, refuse CPU tensors, and carry fake-tensor shape rules."""
import numpy as np
import pytest
import torch

from dsic_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g) * 2 - 1) * scale).cuda()


def test_custom_ops_match_the_direct_calls():
    from dsic_amd import entropy, ops, torch_ops  # noqa: F401  (registers the ops)
    x = _rand((2, 64, 64, 128), 1)
    b = _rand((128,), 2, 0.5)
    beta, gamma = _rand((128,), 3, 0.2) + 0.8, _rand((128,), 4, 0.05) + 0.1
    w = _rand((128, 128, 3, 3), 5, 0.05)
    u = ops.split_wino_weight_bf16(ops.pack_wino_weight(w), 128, 128)
    y = torch.ops.dsic.conv3x3_wino(x, u, b, beta, gamma, 128, ops.ACT_GDN, False, False)
    assert torch.equal(y, ops.conv3x3_wino_nhwc(x, u, b, 128, ops.ACT_GDN, beta, gamma))
    wp = ops.pack_conv_weight(w)
    y2 = torch.ops.dsic.conv2d_bias_act(x, wp, b, None, None, 128, 3, 1, ops.ACT_RELU)
    assert torch.equal(y2, ops.conv2d_nhwc(x, wp, b, 128, 3, 1, ops.ACT_RELU))
    wt = _rand((128, 128, 5, 5), 6, 0.05)
    u4 = ops.split_wino_weight_bf16(ops.pack_wino_convT_weight(wt), 128, 128, 4)
    xt = _rand((2, 32, 32, 128), 7)
    y3 = torch.ops.dsic.convT5s2_bias_act(xt, u4, b, beta, gamma, 128, ops.ACT_IGDN)
    assert y3.shape == (2, 64, 64, 128)
    assert torch.equal(y3, ops.conv_transpose2d_wino_nhwc(xt, u4, b, 128, ops.ACT_IGDN, beta, gamma))
    # rate + coder on small latents
    yl = (_rand((2, 4, 6, 192), 8, 6.0)).contiguous()
    zl = (_rand((2, 1, 2, 128), 9, 6.0)).contiguous()
    sigma = _rand((2, 192), 10, 0.5) + 1.5
    nu = _rand((2, 192), 11, 1.0) + 4.0
    zls = _rand((128,), 12, 0.2)
    r = torch.ops.dsic.rate(yl, zl, sigma, nu, zls)
    d = ops.rate(yl, zl, sigma, nu, zls)
    for got, key in zip(r, ("y_hat_nhwc", "y_tilde", "z_tilde", "nll_y", "nll_z", "sums")):
        assert torch.equal(got, d[key]), key
    sz = torch.exp(zls)
    enc = torch.ops.dsic.range_encode(r[1], r[2], sigma, nu, sz, 10, 64)
    c = entropy.compress_latents(d["y_tilde"], d["z_tilde"], sigma, nu, sz, 10, 64)
    assert torch.equal(enc[0], c["bytes"]) and torch.equal(enc[1], c["lengths"]) and int(enc[3].item()) == 0
    # no CPU kernel behind the dispatcher key
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.dsic.conv3x3_wino(x.cpu(), u.cpu(), b.cpu(), None, None, 128, 0, False, False)
    # fake-tensor shape rules
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty((2, 64, 64, 128), device="cuda")
        fu = torch.empty(u.shape, dtype=torch.uint8, device="cuda")
        fb = torch.empty((128,), device="cuda")
        assert torch.ops.dsic.conv3x3_wino(fx, fu, fb, None, None, 128, 0, False, True).shape == (2, 32, 32, 512)
        assert torch.ops.dsic.convT5s2_bias_act(fx, fu, fb, None, None, 128, 0).shape == (2, 128, 128, 128)
