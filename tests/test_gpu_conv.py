"""HIP conv kernels (through the C ABI) vs the oracle ops on the same inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dsic_amd import ops as _ops
    return _ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _pad8(x_nhwc):
    c = x_nhwc.shape[-1]
    if c % 8 == 0:
        return x_nhwc
    return torch.nn.functional.pad(x_nhwc, (0, 8 - c % 8))


def _tol(ref, K):
    return 2e-6 * float(ref.abs().max()) * max(1.0, K ** 0.5 / 8) + 1e-6


def test_layout_roundtrip(ops):
    x = _rand((3, 5, 7, 11), 0).cuda()
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y), x)
    img = _rand((2, 3, 9, 13), 1).cuda()
    p = ops.image_to_nhwc8(img)
    assert torch.equal(p[..., :3], img.permute(0, 2, 3, 1))
    assert float(p[..., 3:].abs().max()) == 0.0


def test_units_against_reference_fixtures(ops):
    """Reference conv()/ConvTranspose2d outputs recorded by make_golden.py."""
    u = np.load(os.path.join(GOLDEN, "units.npz"))
    for tag in sorted({k.split("/")[0] for k in u.files if k.startswith("conv_k")}):
        k = int(tag.split("_")[1][1])
        s = int(tag.split("_")[1][3])
        if k == 1:
            continue
        w = torch.from_numpy(u[tag + "/w"]).cuda()
        b = torch.from_numpy(u[tag + "/b"]).cuda()
        x = torch.from_numpy(u[tag + "/x"]).cuda()
        y = ops.conv2d_nhwc(_nhwc(x), ops.pack_conv_weight(w), b, w.shape[0], k, s)
        ref = torch.from_numpy(u[tag + "/y"])
        np.testing.assert_allclose(ops.nhwc_to_nchw(y).cpu().numpy(), ref.numpy(), atol=_tol(ref, 200))
    for tag in sorted({k.split("/")[0] for k in u.files if k.startswith("convT")}):
        w = torch.from_numpy(u[tag + "/w"]).cuda()
        b = torch.from_numpy(u[tag + "/b"]).cuda()
        x = torch.from_numpy(u[tag + "/x"]).cuda()
        ref = torch.from_numpy(u[tag + "/y"])
        cout = w.shape[1]
        if cout % 8 == 0:
            y = ops.nhwc_to_nchw(ops.conv_transpose2d_nhwc(_nhwc(x), ops.pack_convT_weight(w), b, cout))
        else:
            y = ops.conv_transpose2d_image(_nhwc(x), ops.pack_convT_image_weight(w), b, cout)
            # the image layer runs on split-bf16 MFMAs (2 planes) unless DSIC_WINO_BF16=0: 4x the fp32 class
            np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=_tol(ref, 200) * 4)
            continue
        np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=_tol(ref, 200))


CONV_CASES = [
    # B, Cin, Cout, H, W, k, s, act
    (1, 3, 128, 32, 48, 3, 1, "gdn"),        # g_a.0 shape family (Cin padded to 8)
    (2, 128, 128, 40, 24, 5, 2, "gdn"),      # g_a.2/6/10
    (1, 128, 128, 19, 37, 3, 1, "gdn"),      # odd sizes, tile borders
    (1, 128, 128, 17, 33, 5, 2, "none"),
    (3, 128, 192, 10, 14, 5, 2, "none"),     # g_a.14 (two column tiles per wave)
    (3, 192, 128, 6, 8, 3, 1, "relu"),       # h_a.0, 8x8x2 tile
    (5, 128, 128, 6, 8, 5, 2, "relu"),       # h_a.4 -> 3x4 grid, 4x4x8 tile, ragged batch
    (9, 128, 128, 3, 4, 5, 2, "none"),       # h_a.6
    (2, 128, 128, 16, 16, 3, 1, "igdn"),     # g_s conv
    (1, 4, 128, 16, 24, 3, 1, "gdn"),        # 4-band first layer
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,act", CONV_CASES)
def test_conv2d_vs_oracle(ops, B, Cin, Cout, H, W, k, s, act):
    x = _rand((B, Cin, H, W), 1, 2.0)
    w = _rand((Cout, Cin, k, k), 2, (Cin * k * k) ** -0.5 * 2)
    b = _rand((Cout,), 3, 0.5)
    beta_p = torch.sqrt(0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 2 ** -18)
    gam_p = torch.sqrt(0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 2 ** -18)
    sd = {"p.weight": w, "p.bias": b}
    ref = O._conv(sd, "p", x, s)
    code = {"none": ops.ACT_NONE, "gdn": ops.ACT_GDN, "igdn": ops.ACT_IGDN, "relu": ops.ACT_RELU}[act]
    if act in ("gdn", "igdn"):
        ref = O.gdn(ref, beta_p, gam_p.view(-1, 1, 1, 1), act == "igdn")
    elif act == "relu":
        ref = torch.relu(ref)
    beta = (beta_p ** 2 - 2 ** -18).cuda()
    gamma = (gam_p ** 2 - 2 ** -18).cuda()
    y = ops.conv2d_nhwc(_pad8(_nhwc(x)).cuda(), ops.pack_conv_weight(w.cuda()), b.cuda(), Cout, k, s,
                        code, beta, gamma)
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    assert err <= _tol(ref, Cin * k * k) * 4, (err, float(ref.abs().max()))


CONVT_CASES = [
    (2, 192, 128, 5, 7, "igdn"),     # g_s.0
    (1, 128, 128, 16, 24, "igdn"),   # g_s.4/8
    (1, 128, 128, 9, 17, "relu"),    # h_s, odd grid
    (9, 128, 128, 2, 3, "relu"),     # h_s.0 on tiny z, ragged 4x4x8 tile
    (3, 128, 128, 8, 6, "none"),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,act", CONVT_CASES)
def test_conv_transpose_vs_oracle(ops, B, Cin, Cout, H, W, act):
    x = _rand((B, Cin, H, W), 11, 2.0)
    w = _rand((Cin, Cout, 5, 5), 12, (Cin * 6.25) ** -0.5 * 2)
    b = _rand((Cout,), 13, 0.5)
    beta_p = torch.sqrt(0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 2 ** -18)
    gam_p = torch.sqrt(0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 2 ** -18)
    ref = O._convT({"p.weight": w, "p.bias": b}, "p", x)
    code = {"none": ops.ACT_NONE, "igdn": ops.ACT_IGDN, "relu": ops.ACT_RELU}[act]
    if act == "igdn":
        ref = O.gdn(ref, beta_p, gam_p.view(-1, 1, 1, 1), True)
    elif act == "relu":
        ref = torch.relu(ref)
    y = ops.conv_transpose2d_nhwc(_nhwc(x).cuda(), ops.pack_convT_weight(w.cuda()), b.cuda(), Cout, code,
                                  (beta_p ** 2 - 2 ** -18).cuda(), (gam_p ** 2 - 2 ** -18).cuda())
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    assert err <= _tol(ref, Cin * 9) * 4, (err, float(ref.abs().max()))


@pytest.mark.parametrize("B,Cin,Cimg,H,W", [(2, 128, 3, 16, 24), (1, 128, 4, 9, 5), (9, 128, 3, 3, 4),
                                             (1, 128, 3, 40, 70), (1, 64, 3, 17, 33), (2, 32, 1, 33, 65)])
def test_conv_transpose_image_vs_oracle(ops, B, Cin, Cimg, H, W):
    x = _rand((B, Cin, H, W), 21, 2.0)
    w = _rand((Cin, Cimg, 5, 5), 22, (Cin * 6.25) ** -0.5 * 2)
    b = _rand((Cimg,), 23, 0.5)
    ref = O._convT({"p.weight": w, "p.bias": b}, "p", x)
    got = ops.conv_transpose2d_image(_nhwc(x).cuda(), ops.pack_convT_image_weight(w.cuda()), b.cuda(), Cimg).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    print(f"image layer err/|ref|max = {err / float(ref.abs().max()):.2e}")
    # split-bf16 contraction (2 planes, 3 products) unless DSIC_WINO_BF16=0: dropped cross terms 2^-16 per product
    import os
    assert err <= _tol(ref, Cin * 9) * (4 if os.environ.get("DSIC_WINO_BF16") == "0" else 16), (err, float(ref.abs().max()))


def test_bad_arguments_raise(ops):
    x = torch.zeros((1, 4, 4, 8), device="cuda")
    w = torch.zeros(9 * 32 * 8, device="cuda")
    b = torch.zeros(8, device="cuda")
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(x, w, b, 8, 3, 2)          # (k,stride) not supported
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(x, w, b, 8, 3, 1, ops.ACT_GDN)   # GDN without beta/gamma
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x.cpu(), w, b, 8, 3, 1)    # no CPU fallback


@pytest.mark.parametrize("B,C,Cout,H,W,act", [(2, 3, 128, 40, 56, "gdn"), (1, 4, 128, 17, 33, "gdn"),
                                              (1, 3, 64, 8, 16, "none"), (3, 3, 128, 9, 5, "relu")])
def test_first_layer_kernel_vs_oracle(ops, B, C, Cout, H, W, act):
    x = _rand((B, C, H, W), 31, 1.0)
    w = _rand((Cout, C, 3, 3), 32, (C * 9) ** -0.5 * 2)
    b = _rand((Cout,), 33, 0.5)
    beta_p = torch.sqrt(0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 2 ** -18)
    gam_p = torch.sqrt(0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 2 ** -18)
    ref = O._conv({"p.weight": w, "p.bias": b}, "p", x, 1)
    code = {"none": ops.ACT_NONE, "gdn": ops.ACT_GDN, "relu": ops.ACT_RELU}[act]
    if act == "gdn":
        ref = O.gdn(ref, beta_p, gam_p.view(-1, 1, 1, 1), False)
    elif act == "relu":
        ref = torch.relu(ref)
    y = ops.conv_first_nchw(x.cuda(), w.cuda(), b.cuda(), code, (beta_p ** 2 - 2 ** -18).cuda(),
                            (gam_p ** 2 - 2 ** -18).cuda())
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    # (the layer runs on split-bf16 MFMAs unless DSIC_WINO_BF16=0: the tolerance class of the other split layers)
    from dsic_amd import layers as Lm
    assert float((got - ref).abs().max()) <= _tol(ref, C * 9) * 4 * (4.0 if Lm.WINO_BF16 else 1.0)


WINO_CASES = [
    (1, 128, 128, 16, 32, "gdn"),     # exact tiles
    (2, 128, 128, 19, 37, "igdn"),    # odd sizes: partial Winograd tiles at the borders
    (1, 192, 128, 16, 16, "relu"),    # h_a.0: 6 chunks
    (3, 128, 128, 6, 10, "none"),     # fewer pixels than one tile
    (2, 64, 64, 24, 40, "gdn"),       # half-width column tiles idle
    (70, 32, 128, 8, 16, "none"),     # one tile per workgroup, single chunk
    (300, 32, 64, 8, 16, "relu"),     # single-chunk layer, more tiles than workgroups (ticket loop)
    (4, 128, 128, 96, 160, "gdn"),    # 480 tiles: every workgroup takes several through the ticket
    # H, W multiples of 16 and >= 16 work items per image: the 64-tile two-pass kernel (conv_wino_bf16m.hip)
    (2, 128, 128, 64, 64, "gdn"),     # g_a.8 / g_s.6
    (1, 128, 64, 64, 96, "igdn"),     # half of the channel groups idle, H != W
    (2, 192, 128, 64, 64, "relu"),    # 12 chunks per pass
    (3, 64, 128, 64, 80, "none"),     # 4 chunks per pass: the shortest pipeline
    (5, 128, 128, 128, 128, "gdn"),   # g_a.4 / g_s.10: 320 work items, every workgroup takes a second one
]


def _wino_weights(ops, variant, u, Cout, Cin, nphase=1):
    """variant "fp32": the fp32-input MFMA kernel (conv_wino.hip); "bf16": operands split into bf16
    planes, bf16 MFMA with fp32 accumulation (conv_wino_bf16.hip; needs Cin >= 64)."""
    if variant == "fp32":
        return u
    if Cin < 64:
        pytest.skip("the split-bf16 kernel needs at least four 16-channel chunks")
    return ops.split_wino_weight_bf16(u, Cout, Cin, nphase)


# Dropped cross terms of the two-plane split: relative 2^-16 per product, i.e. up to ~16x the fp32
# rounding of one product; summed over K products with random signs the output error stays within
# 4x the fp32 tolerance class used above (measured: see DESIGN.md section 3).
_BF16_TOL = {"fp32": 1.0, "bf16": 4.0}


@pytest.mark.parametrize("variant", ["fp32", "bf16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,act", WINO_CASES)
def test_winograd_conv_vs_oracle(ops, B, Cin, Cout, H, W, act, variant):
    x = _rand((B, Cin, H, W), 41, 2.0)
    w = _rand((Cout, Cin, 3, 3), 42, (Cin * 9) ** -0.5 * 2)
    b = _rand((Cout,), 43, 0.5)
    beta_p = torch.sqrt(0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 2 ** -18)
    gam_p = torch.sqrt(0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 2 ** -18)
    ref = O._conv({"p.weight": w, "p.bias": b}, "p", x, 1)
    code = {"none": ops.ACT_NONE, "gdn": ops.ACT_GDN, "igdn": ops.ACT_IGDN, "relu": ops.ACT_RELU}[act]
    if act in ("gdn", "igdn"):
        ref = O.gdn(ref, beta_p, gam_p.view(-1, 1, 1, 1), act == "igdn")
    elif act == "relu":
        ref = torch.relu(ref)
    u = _wino_weights(ops, variant, ops.pack_wino_weight(w.cuda()), Cout, Cin)
    y = ops.conv3x3_wino_nhwc(_nhwc(x).cuda(), u, b.cuda(), Cout, code,
                              (beta_p ** 2 - 2 ** -18).cuda(), (gam_p ** 2 - 2 ** -18).cuda())
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    print(f"[{variant}] err/|ref|max = {err / float(ref.abs().max()):.2e}")
    # Winograd reorders the fp32 sums (transform adds before the products): same tolerance class
    assert err <= _tol(ref, Cin * 9) * 6 * _BF16_TOL[variant], (err, float(ref.abs().max()))
    # and against float64 the error stays at fp32 level
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if act == "none":
        assert float((got.double() - ref64).abs().max()) < 2e-5 * _BF16_TOL[variant] * float(ref64.abs().max())


@pytest.mark.parametrize("B,Cs,Cout,H,W,act", [(1, 128, 128, 32, 64, "gdn"), (2, 128, 128, 20, 36, "none"),
                                               (1, 64, 64, 16, 16, "relu"), (3, 128, 128, 6, 10, "gdn"),
                                               (2, 128, 128, 192, 224, "gdn"),
                                               # outputs of 64x64 / 64x80 / 128x128: the 64-tile two-pass kernel
                                               (2, 128, 128, 128, 128, "gdn"), (1, 64, 64, 128, 160, "none"),
                                               (3, 128, 128, 256, 256, "relu")])
@pytest.mark.parametrize("variant", ["fp32", "bf16"])
def test_conv5x5_stride2_as_winograd_over_space_to_depth(ops, B, Cs, Cout, H, W, act, variant):
    """conv(C,C,5,2) == 3x3 Winograd over the space-to-depth input with 4*C channels."""
    x = _rand((B, Cs, H, W), 51, 2.0)
    w = _rand((Cout, Cs, 5, 5), 52, (Cs * 25) ** -0.5 * 2)
    b = _rand((Cout,), 53, 0.5)
    beta_p = torch.sqrt(0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 2 ** -18)
    gam_p = torch.sqrt(0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 2 ** -18)
    ref = O._conv({"p.weight": w, "p.bias": b}, "p", x, 2)
    code = {"none": ops.ACT_NONE, "gdn": ops.ACT_GDN, "relu": ops.ACT_RELU}[act]
    if act == "gdn":
        ref = O.gdn(ref, beta_p, gam_p.view(-1, 1, 1, 1), False)
    elif act == "relu":
        ref = torch.relu(ref)
    xs = ops.space_to_depth(_nhwc(x).cuda())
    u = _wino_weights(ops, variant, ops.pack_wino_s2_weight(w.cuda()), Cout, 4 * Cs)
    y = ops.conv3x3_wino_nhwc(xs, u, b.cuda(), Cout, code,
                              (beta_p ** 2 - 2 ** -18).cuda(), (gam_p ** 2 - 2 ** -18).cuda(), s2d_in=True)
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    print(f"[{variant}] err/|ref|max = {err / float(ref.abs().max()):.2e}")
    assert err <= _tol(ref, Cs * 25) * 6 * _BF16_TOL[variant], (err, float(ref.abs().max()))


def test_space_to_depth_epilogues(ops):
    """The s2d_out stores of the first-layer and Winograd kernels equal space_to_depth(normal output)."""
    x = _rand((2, 3, 16, 32), 61, 1.0).cuda()
    w = _rand((128, 3, 3, 3), 62, 0.4).cuda()
    b = _rand((128,), 63, 0.5).cuda()
    a = ops.conv_first_nchw(x, w, b)
    assert torch.equal(ops.conv_first_nchw(x, w, b, s2d_out=True), ops.space_to_depth(a))
    assert torch.equal(ops.depth_to_space(ops.space_to_depth(a)), a)
    w2 = ops.pack_wino_weight(_rand((128, 128, 3, 3), 64, 0.05).cuda())
    y = ops.conv3x3_wino_nhwc(a, w2, b, 128)
    assert torch.equal(ops.conv3x3_wino_nhwc(a, w2, b, 128, s2d_out=True), ops.space_to_depth(y))


def test_two_pass_kernel_stores(ops):
    """The 64-tile two-pass kernel stores straight from the accumulators: the space-to-depth order and a
    channel slice of a wider tensor equal the plain output; a second launch gives the same bits."""
    from dsic_amd import lib
    L = lib.load()
    assert L.dsic_wino_bf16_m64(64, 96, 128, 1) == 1 and L.dsic_wino_bf16_m64(64, 90, 128, 1) == 0
    assert L.dsic_wino_bf16_m64(16, 16, 128, 4) == 1 and L.dsic_wino_bf16_m64(16, 16, 128, 1) == 0 and L.dsic_wino_bf16_m64(32, 32, 128, 1) == 1
    x = _rand((2, 64, 96, 128), 65, 1.0).cuda()
    b = _rand((128,), 66, 0.5).cuda()
    beta = (0.5 + torch.rand(128, generator=torch.Generator().manual_seed(4))).cuda()
    gamma = (0.02 + 0.28 * torch.rand(128, generator=torch.Generator().manual_seed(5))).cuda()
    u = ops.split_wino_weight_bf16(ops.pack_wino_weight(_rand((128, 128, 3, 3), 67, 0.05).cuda()), 128, 128)
    y = ops.conv3x3_wino_nhwc(x, u, b, 128, ops.ACT_GDN, beta, gamma)
    assert torch.equal(ops.conv3x3_wino_nhwc(x, u, b, 128, ops.ACT_GDN, beta, gamma), y)
    assert torch.equal(ops.conv3x3_wino_nhwc(x, u, b, 128, ops.ACT_GDN, beta, gamma, s2d_out=True), ops.space_to_depth(y))
    u64 = ops.split_wino_weight_bf16(ops.pack_wino_weight(_rand((64, 128, 3, 3), 68, 0.05).cuda()), 64, 128)
    y64 = ops.conv3x3_wino_nhwc(x, u64, b[:64].contiguous(), 64, ops.ACT_RELU)
    wide = torch.full((2, 64, 96, 192), -7.0, device="cuda")
    ops.conv3x3_wino_nhwc(x, u64, b[:64].contiguous(), 64, ops.ACT_RELU, out=wide, out_coff=64)
    assert torch.equal(wide[..., 64:128], y64)
    assert float(wide[..., :64].max()) == -7.0 and float(wide[..., 128:].max()) == -7.0


@pytest.mark.parametrize("B,Cin,Cout,H,W,act", [(2, 192, 128, 5, 7, "igdn"), (1, 128, 128, 16, 24, "igdn"),
                                                (1, 128, 128, 9, 17, "relu"), (9, 128, 128, 2, 3, "none"),
                                                (2, 128, 64, 32, 16, "none"), (3, 128, 128, 40, 72, "igdn"),
                                                # 16 and more (tile, phase) items per image: the 64-tile two-pass kernel
                                                (2, 128, 128, 32, 32, "igdn"), (1, 128, 64, 32, 48, "none"),
                                                (1, 192, 128, 32, 32, "relu"), (3, 128, 128, 64, 64, "igdn")])
@pytest.mark.parametrize("variant", ["fp32", "bf16"])
def test_conv_transpose_winograd_vs_oracle(ops, B, Cin, Cout, H, W, act, variant):
    x = _rand((B, Cin, H, W), 71, 2.0)
    w = _rand((Cin, Cout, 5, 5), 72, (Cin * 6.25) ** -0.5 * 2)
    b = _rand((Cout,), 73, 0.5)
    beta_p = torch.sqrt(0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4)) + 2 ** -18)
    gam_p = torch.sqrt(0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 2 ** -18)
    ref = O._convT({"p.weight": w, "p.bias": b}, "p", x)
    code = {"none": ops.ACT_NONE, "igdn": ops.ACT_IGDN, "relu": ops.ACT_RELU}[act]
    if act == "igdn":
        ref = O.gdn(ref, beta_p, gam_p.view(-1, 1, 1, 1), True)
    elif act == "relu":
        ref = torch.relu(ref)
    u = _wino_weights(ops, variant, ops.pack_wino_convT_weight(w.cuda()), Cout, Cin, 4)
    y = ops.conv_transpose2d_wino_nhwc(_nhwc(x).cuda(), u, b.cuda(), Cout, code,
                                       (beta_p ** 2 - 2 ** -18).cuda(), (gam_p ** 2 - 2 ** -18).cuda())
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    print(f"[{variant}] err/|ref|max = {err / float(ref.abs().max()):.2e}")
    assert err <= _tol(ref, Cin * 9) * 6 * _BF16_TOL[variant], (err, float(ref.abs().max()))


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 20, 36)])
def test_conv5x5_stride2_192_channels_as_two_winograd_slices(ops, B, H, W):
    """g_a.14, conv(128, 192, 5, 2) (layers.py:72): 192 output channels run as a 128- and a 64-channel
    call of the split-bf16 Winograd kernel into one [B,H/2,W/2,192] tensor (out_cstride / out_coff)."""
    from dsic_amd import layers as Lm
    if not Lm.WINO_BF16:
        pytest.skip("channel slices exist in the split-bf16 kernel only")
    x = _rand((B, 128, H, W), 81, 2.0)
    m = Lm.Conv2d(128, 192, 5, 2)
    w = _rand((192, 128, 5, 5), 82, (128 * 25) ** -0.5 * 2)
    b = _rand((192,), 83, 0.5)
    with torch.no_grad():
        m.weight.copy_(w)
        m.bias.copy_(b)
    m = m.cuda()
    assert m.use_winograd_s2
    ref = O._conv({"p.weight": w, "p.bias": b}, "p", x, 2)
    xs = ops.space_to_depth(_nhwc(x).cuda())
    y = m.run_nhwc(xs, x_is_s2d=True)
    got = ops.nhwc_to_nchw(y).cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    assert err <= _tol(ref, 128 * 25) * 6 * _BF16_TOL["bf16"], (err, float(ref.abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,H,W,act,s2d_in,s2d_out", [
    (5, 128, 128, 16, 16, "gdn", False, True),     # g_a.12: 2 tiles per image -> 2 work items per tile, s2d store
    (3, 512, 128, 8, 8, "none", True, False),      # g_a.14 slice / h_a: 1 tile per image -> 4 work items
    (70, 512, 128, 16, 16, "gdn", True, False),    # g_a.10 shape, more work items than workgroups
    (2, 192, 128, 16, 16, "relu", False, False),   # h_a.0: 12 chunks -> runs of 6
    (1, 128, 64, 6, 10, "igdn", False, False),
])
def test_winograd_split_k_equals_unsplit_sums(ops, B, Cin, Cout, H, W, act, s2d_in, s2d_out):
    """Layers with fewer than four tiles per image share a tile's input channels between work items
    (dsic_wino_bf16_ksplit) and add the partial sums in a second launch: same result as the unsplit kernel up
    to the fp32 rounding of the regrouped sum, same addressing (space-to-depth store, channel slices)."""
    from dsic_amd import lib, ops as OPS
    S = lib.load().dsic_wino_bf16_ksplit(H, W, Cin)
    assert S > 1 and (Cin // 16) % S == 0
    x = _rand((B, H, W, Cin), 91, 2.0).cuda()
    if s2d_in:
        w = _rand((Cout, Cin // 4, 5, 5), 92, (Cin // 4 * 25) ** -0.5 * 2).cuda()
        u = ops.split_wino_weight_bf16(ops.pack_wino_s2_weight(w), Cout, Cin, 1)
    else:
        w = _rand((Cout, Cin, 3, 3), 92, (Cin * 9) ** -0.5 * 2).cuda()
        u = ops.split_wino_weight_bf16(ops.pack_wino_weight(w), Cout, Cin, 1)
    b = _rand((Cout,), 93, 0.5).cuda()
    beta = (0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(4))).cuda()
    gamma = (0.02 + 0.28 * torch.rand(Cout, generator=torch.Generator().manual_seed(5))).cuda()
    code = {"none": ops.ACT_NONE, "gdn": ops.ACT_GDN, "igdn": ops.ACT_IGDN, "relu": ops.ACT_RELU}[act]
    saved = OPS.WINO_SPLITK
    try:
        OPS.WINO_SPLITK = False
        ref = ops.conv3x3_wino_nhwc(x, u, b, Cout, code, beta, gamma, s2d_in=s2d_in, s2d_out=s2d_out)
        OPS.WINO_SPLITK = True
        got = ops.conv3x3_wino_nhwc(x, u, b, Cout, code, beta, gamma, s2d_in=s2d_in, s2d_out=s2d_out)
        again = ops.conv3x3_wino_nhwc(x, u, b, Cout, code, beta, gamma, s2d_in=s2d_in, s2d_out=s2d_out)
        # a Cout slice of a wider tensor: the partial buffers use the same stride and offset
        wide = torch.full((B, H, W, Cout + 64), -7.0, device="cuda")
        if not s2d_out:
            ops.conv3x3_wino_nhwc(x, u, b, Cout, code, beta, gamma, s2d_in=s2d_in, out=wide, out_coff=64)
    finally:
        OPS.WINO_SPLITK = saved
    assert torch.equal(got, again)                                  # fixed summation order
    assert got.shape == ref.shape
    err = float((got - ref).abs().max())
    assert 0 < err <= 4e-6 * float(ref.abs().max()), err            # regrouped fp32 sums, nothing else
    if not s2d_out:
        assert torch.equal(wide[..., 64:], got) and bool((wide[..., :64] == -7.0).all())
    # one image alone takes the same split: batch invariance
    OPS.WINO_SPLITK = True
    solo = ops.conv3x3_wino_nhwc(x[B - 1:].contiguous(), u, b, Cout, code, beta, gamma, s2d_in=s2d_in, s2d_out=s2d_out)
    assert torch.equal(solo[0], got[B - 1])


def test_variant_switch_is_one_run_time_call(ops):
    """dsic_set_split_bf16 (layers.set_wino_bf16): ONE switch through the C ABI moves the first layer, the Winograd
    layers and the image layer between the split-bf16 and the fp32-input MFMA kernels; layers built before the switch
    follow it (their packed weights are cached per variant)."""
    from dsic_amd import layers as Lm, lib
    L = lib.load()
    before = Lm.wino_bf16()
    torch.manual_seed(5)
    g_a = Lm.AnalysisTransform(128, 192, 3).cuda()
    g_s = Lm.SynthesisTransform(128, 192, 3).cuda()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    try:
        outs = {}
        for variant in (True, False, True):
            Lm.set_wino_bf16(variant)
            assert bool(L.dsic_split_bf16()) == variant and Lm.WINO_BF16 == variant
            y = g_a(x)
            xh = g_s(torch.round(y))
            outs.setdefault(variant, []).append((y.clone(), xh.clone()))
        (y1, x1), (y1b, x1b) = outs[True]
        (y0, x0), = outs[False]
        assert torch.equal(y1, y1b) and torch.equal(x1, x1b)          # back on the first variant: the same bits
        assert not torch.equal(y1, y0) and not torch.equal(x1, x0)    # the switch reached the kernels
        scale = float(y0.abs().max())
        assert float((y1 - y0).abs().max()) <= 2e-4 * scale           # two fp32-class evaluations of the same transform
    finally:
        Lm.set_wino_bf16(before)
