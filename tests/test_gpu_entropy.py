"""GPU entropy path (tables, range encoder/decoder) vs the oracle: bit-exact on the same inputs."""
import numpy as np
import pytest
import torch

from dsic_amd import synthetic as S
from oracle import entropy_ref as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from dsic_amd.model import CompressionModel
    sd = S.make_state_dict(seed=1)
    m = CompressionModel(N=128, M=192, spatial_params=False, min_nu=2, max_nu=100.0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.cuda().eval()


def _oracle_inputs(out, model):
    y = out["y_tilde"].cpu().numpy()
    z = out["z_tilde"].cpu().numpy()
    sy = out["sigma"][:, :, 0, 0].cpu().numpy()
    ny = out["nu"][:, :, 0, 0].cpu().numpy()
    from dsic_amd import entropy
    sz = entropy.sigma_z_of(model).cpu().numpy()
    return y, z, sy, ny, sz


@pytest.mark.parametrize("B,H,W", [(2, 128, 96), (1, 256, 256), (3, 48, 80)])
def test_custom_compress_is_bit_exact_to_oracle(model, B, H, W):
    from dsic_amd import entropy
    x = torch.from_numpy(S.make_patches(200, B, H, W)).cuda()
    out = model(x, quant_mode="round")
    y, z, sy, ny, sz = _oracle_inputs(out, model)
    want = E.compress(y, z, sy, ny, sz, tail=10)
    got = entropy.custom_compress(model, x, tail=10)
    # the reference's keys (eval_selfcontained_entropy.py:68-74) plus the encoder's numerics tag
    assert set(got) == {"strings", "shape_y", "shape_z", "min_y", "max_y", "min_z", "max_z", "numerics"}
    for k in ("shape_y", "shape_z", "min_y", "max_y", "min_z", "max_z"):
        assert got[k] == want[k], k
    for b in range(B):
        assert got["strings"][b][0] == want["strings"][b][0], f"z string of image {b}"
        assert got["strings"][b][1] == want["strings"][b][1], f"y string of image {b}"
    # the coded size tracks the density estimate (model.py:76) from above, within a few percent
    est_bits = float(out.sums.sum().item())
    real_bits = 8 * sum(len(s) for e in got["strings"] for s in e)
    assert 0.98 * est_bits < real_bits < 1.10 * est_bits + 64 * B, (est_bits, real_bits)
    assert abs(entropy.real_bpp(got, H, W) - real_bits / (H * W)) < 1e-12
    # decode on the GPU: identical latents -> identical reconstruction
    x_hat = entropy.custom_decompress(model, got)
    assert torch.equal(x_hat, out["x_hat"].clamp(0, 1))
    # and the oracle decodes the GPU's strings
    for b in range(B):
        assert np.array_equal(E.decode_z(got, b, sz), z[b])
        assert np.array_equal(E.decode_y(got, b, sy[b], ny[b]), y[b])


def test_device_tables_equal_oracle(model):
    from dsic_amd import entropy
    x = torch.from_numpy(S.make_patches(300, 2, 64, 64)).cuda()
    out = model(x, quant_mode="round")
    y, z, sy, ny, sz = _oracle_inputs(out, model)
    meta = entropy.latent_support(out["y_tilde"], out["z_tilde"], 10)
    m = meta.cpu().numpy()
    for b in range(2):
        assert m[b, 0] == int(y[b].min()) - 10 and m[b, 1] == int(y[b].max() - y[b].min()) + 21
        assert m[b, 2] == int(z[b].min()) - 10 and m[b, 3] == int(z[b].max() - z[b].min()) + 21
    tab_y, tab_z, err = entropy.cdf_tables(out["sigma"][:, :, 0, 0].contiguous(), out["nu"][:, :, 0, 0].contiguous(),
                                           entropy.sigma_z_of(model), meta, 128)
    assert int(err.item()) == 0
    ty, tz = tab_y.cpu().numpy(), tab_z.cpu().numpy()
    for b in range(2):
        Ly, Lz = int(m[b, 1]), int(m[b, 3])
        assert np.array_equal(ty[b, :, :Ly], E.tables_student(sy[b], ny[b], int(m[b, 0]), Ly))
        assert np.array_equal(tz[b, :, :Lz], E.tables_gauss(sz, int(m[b, 2]), Lz))


def test_device_z_tables_vs_reference_fixture():
    """The device tables_kernel on the reference's own z cases (tests/golden/entropy_ref.npz, made by
    running eval_selfcontained_entropy.py:36-48): identical to the oracle, and equal to the reference's
    uint16 tables (after the documented spreading) except for erf's last bit: max |diff| 1 on < 0.1 %."""
    import os
    from dsic_amd import entropy
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "entropy_ref.npz"))
    total = bad = 0
    for i in list(range(0, int(g["zsweep/count"][0]), 3)) + [35]:
        sig, zt, ref = g[f"zsweep/{i}/sigma_z"], g[f"zsweep/{i}/z_tilde"], g[f"zsweep/{i}/cdf_u16"]
        zmin, Ls = int(zt.min()) - 10, ref.shape[0] - 1
        Lmax = (Ls + 7) // 8 * 8
        meta = torch.tensor([[0, 1, zmin, Ls]], dtype=torch.int32, device="cuda")
        tab_y, tab_z, err = entropy.cdf_tables(torch.ones((1, 1), device="cuda"), torch.full((1, 1), 3.0, device="cuda"),
                                               torch.from_numpy(sig).cuda(), meta, Lmax)
        assert int(err.item()) == 0
        got = tab_z[0, :, :Ls].cpu().numpy().astype(np.int64)
        assert np.array_equal(got, E.tables_gauss(sig, zmin, Ls))
        k = np.arange(Ls)[None, :]
        want = ref[:Ls].T.astype(np.int64) * (65536 - Ls) // 65535 + k          # spreading of DESIGN.md §4
        b = (np.float32(zmin) - np.float32(0.5)) / sig
        e = (np.float32(zmin + Ls) - np.float32(0.5)) / sig
        from scipy import special
        ok = 0.5 * (special.erf(e / np.sqrt(2.0)) - special.erf(b / np.sqrt(2.0))) > 0.5
        d = np.abs(got - want)[ok]
        if d.size:
            assert d.max() <= 1
        bad += np.count_nonzero(d)
        total += d.size
    assert total > 50000 and bad <= 0.001 * total, (bad, total)


@pytest.mark.parametrize("case", ["random", "peaky", "runs", "single"])
def test_range_coder_stress_vs_oracle(case):
    """Synthetic latents that exercise long pending runs, one-count intervals and tiny streams."""
    from dsic_amd import entropy
    rng = np.random.default_rng(11)
    B, M, N, Hy, Wy, Hz, Wz = 3, 16, 8, 8, 12, 2, 3
    if case == "single":
        B, M, N, Hy, Wy, Hz, Wz = 2, 1, 1, 1, 1, 1, 1
    if case == "random":
        y = np.rint(rng.standard_t(2.5, size=(B, M, Hy, Wy)) * 3).clip(-60, 60)
        sy = rng.uniform(0.5, 6.0, (B, M)); ny = rng.uniform(2.0, 50.0, (B, M))
    elif case == "peaky":   # sigma tiny: almost all mass in one bin, other symbols sit on 1-count intervals
        y = np.zeros((B, M, Hy, Wy)); y[:, :, ::3, ::5] = rng.integers(-9, 10, size=y[:, :, ::3, ::5].shape)
        sy = np.full((B, M), 1e-3); ny = np.full((B, M), 2.0)
    elif case == "runs":    # symmetric two-valued stream around the interval midpoint -> pending runs
        y = np.where(rng.random((B, M, Hy, Wy)) < 0.5, 0.0, -1.0)
        sy = np.full((B, M), 40.0); ny = np.full((B, M), 100.0)
    else:
        y = np.array([3.0, -2.0]).reshape(B, 1, 1, 1)
        sy = np.full((B, M), 1.0); ny = np.full((B, M), 5.0)
    z = np.rint(rng.normal(size=(B, N, Hz, Wz)) * 4)
    sz = rng.uniform(0.5, 5.0, N)
    y, z = y.astype(np.float32), z.astype(np.float32)
    sy, ny, sz = sy.astype(np.float32), ny.astype(np.float32), sz.astype(np.float32)
    want = E.compress(y, z, sy, ny, sz, tail=10)
    c = entropy.compress_latents(torch.from_numpy(y).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(sy).cuda(),
                                 torch.from_numpy(ny).cuda(), torch.from_numpy(sz).cuda(), tail=10, Lmax=192)
    assert int(c["err"].item()) == 0
    lens = c["lengths"].cpu().numpy()
    raw = c["bytes"].cpu().numpy()
    for b in range(B):
        assert raw[b, :lens[b, 0]].tobytes() == want["strings"][b][0]
        assert raw[b, c["cap_z"]:c["cap_z"] + lens[b, 1]].tobytes() == want["strings"][b][1]


def test_errors_are_reported(model):
    from dsic_amd import entropy
    y = torch.zeros((1, 4, 2, 2), device="cuda"); y[0, 0, 0, 0] = 500.0
    z = torch.zeros((1, 2, 1, 1), device="cuda")
    c = entropy.compress_latents(y, z, torch.ones((1, 4), device="cuda"), torch.full((1, 4), 3.0, device="cuda"),
                                 torch.ones(2, device="cuda"), tail=10, Lmax=64)
    with pytest.raises(entropy.EntropyError):
        entropy._check_err(c["err"], "test")


def test_container_round_trip(model):
    from dsic_amd import entropy
    x = torch.from_numpy(S.make_patches(400, 2, 64, 80)).cuda()
    c = entropy.custom_compress(model, x)
    blob = entropy.pack_container(c)
    back = entropy.unpack_container(blob)
    assert back == c
    assert len(blob) == 6 + 4 + 28 + 24 * 2 + sum(len(s) for e in c["strings"] for s in e)
    assert torch.equal(entropy.custom_decompress(model, back), model(x, "round")["x_hat"].clamp(0, 1))
    with pytest.raises(ValueError):
        entropy.unpack_container(blob[:-1])
    with pytest.raises(ValueError):
        entropy.unpack_container(b"nope" + blob)
    # the header carries the encoder's numerics tag (table flow, kernel arithmetic of h_s, ABI version): a stream
    # written by another variant - e.g. DSIC_WINO_BF16=0, whose sigma / nu differ in the last bits - is refused
    # instead of decoding to garbage latents, and so is a container of the rounds before the tag existed
    assert c["numerics"] == entropy.numerics_tag() and (c["numerics"] & 0xFF) == entropy.TABLE_FLOW_VERSION
    other = dict(back, numerics=back["numerics"] ^ (1 << 8))
    with pytest.raises(entropy.EntropyError):
        entropy.custom_decompress(model, other)
    with pytest.raises(entropy.EntropyError):
        entropy.custom_decompress(model, entropy.unpack_container(entropy.pack_container(other)))
    with pytest.raises(ValueError):
        entropy.unpack_container(b"DSIC1\x00" + blob[10:])
    legacy = {k: v for k, v in c.items() if k != "numerics"}     # the reference's own dict (no tag): accepted
    assert torch.equal(entropy.custom_decompress(model, legacy), model(x, "round")["x_hat"].clamp(0, 1))


@pytest.mark.parametrize("spread", [4, 40, 120])
def test_decoder_with_wide_supports(spread):
    """Supports wider than one 64-entry table segment (L up to 261) decode exactly."""
    from dsic_amd import entropy
    rng = np.random.default_rng(spread)
    B, M, N, Hy, Wy = 2, 6, 3, 5, 7
    y = np.rint(rng.normal(size=(B, M, Hy, Wy)) * spread).astype(np.float32)
    z = np.rint(rng.normal(size=(B, N, 2, 2)) * spread).astype(np.float32)
    sy = rng.uniform(0.5 * spread, 1.5 * spread, (B, M)).astype(np.float32)
    ny = rng.uniform(2.0, 30.0, (B, M)).astype(np.float32)
    sz = rng.uniform(0.5 * spread, 1.5 * spread, N).astype(np.float32)
    want = E.compress(y, z, sy, ny, sz, tail=10)
    c = entropy.compress_latents(torch.from_numpy(y).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(sy).cuda(),
                                 torch.from_numpy(ny).cuda(), torch.from_numpy(sz).cuda(), tail=10, Lmax=1000)
    assert int(c["err"].item()) == 0
    lens = c["lengths"].cpu().numpy()
    raw = c["bytes"].cpu().numpy()
    # the wave-parallel table finish (several 64-entry rounds per table here) equals the serial host order
    meta = c["meta"].cpu().numpy()
    ty, tz = c["tab_y"].cpu().numpy().reshape(B, M, -1), c["tab_z"].cpu().numpy().reshape(B, N, -1)
    for b in range(B):
        Ly, Lz = int(meta[b, 1]), int(meta[b, 3])
        assert np.array_equal(ty[b, :, :Ly], E.tables_student(sy[b], ny[b], int(meta[b, 0]), Ly))
        assert np.array_equal(tz[b, :, :Lz], E.tables_gauss(sz, int(meta[b, 2]), Lz))
    for b in range(B):
        assert raw[b, :lens[b, 0]].tobytes() == want["strings"][b][0]
        assert raw[b, c["cap_z"]:c["cap_z"] + lens[b, 1]].tobytes() == want["strings"][b][1]
    # decode the y strings on the GPU with the same tables
    from dsic_amd import lib as _lib
    from dsic_amd.ops import _p, _stream
    y_hat = torch.empty((B, M, Hy, Wy), device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    ybytes = c["bytes"][:, c["cap_z"]:].contiguous()
    ylen = c["lengths"][:, 1].contiguous()
    _lib.check(_lib.load().dsic_range_decode(_p(ybytes), ybytes.shape[1], _p(ylen), 1, 0, _p(c["meta"]), 0,
                                             _p(c["tab_y"]), 1000, B, M, Hy * Wy, 0, _p(y_hat), _p(err), _stream()),
               "range_decode")
    assert int(err.item()) == 0
    assert torch.equal(y_hat.cpu(), torch.from_numpy(y))
    assert int(c["meta"][:, 1].max()) > (64 if spread >= 40 else 0)


def test_spatial_params_model_and_entropy_path():
    """spatial_params=True (layers.py:127-129, model.py:49-51): per-element sigma/nu, a table row per symbol."""
    from dsic_amd import entropy
    from dsic_amd.model import CompressionModel
    from oracle import ref_model as O
    sd = S.make_state_dict(seed=4, spatial_params=True)
    m = CompressionModel(N=128, M=192, spatial_params=True, min_nu=2, max_nu=100.0)
    assert not m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True).missing_keys
    m = m.cuda().eval()
    x = torch.from_numpy(S.make_patches(500, 2, 128, 64))
    out = m(x.cuda(), quant_mode="round")
    ref = O.forward(sd, x, "round")
    assert out["sigma"].shape == ref["sigma"].shape == (2, 192, 8, 4)
    np.testing.assert_allclose(out["sigma"].cpu().numpy(), ref["sigma"].numpy(), rtol=2e-4)
    np.testing.assert_allclose(out["nu"].cpu().numpy(), ref["nu"].numpy(), rtol=2e-4)
    bpp = out.sums.sum(dim=1).cpu().numpy() / (128 * 64)
    bpp_ref = (ref["nll_y"].double().sum(dim=(1, 2, 3)) + ref["nll_z"].double().sum(dim=(1, 2, 3))).numpy() / (128 * 64)
    assert np.max(np.abs(bpp - bpp_ref)) < 1e-4
    ls, ln = m.h_s(out["z_tilde"])
    rls, rln = O.hyper_synthesis(sd, ref["z_tilde"])
    np.testing.assert_allclose(ls.cpu().numpy(), rls.numpy(), rtol=1e-4, atol=1e-5)
    # entropy path on the GPU's own (y, z, sigma, nu): bytes identical to the oracle, exact round trip
    y, z = out["y_tilde"].cpu().numpy(), out["z_tilde"].cpu().numpy()
    sy, ny = out["sigma"].cpu().numpy(), out["nu"].cpu().numpy()
    sz = entropy.sigma_z_of(m).cpu().numpy()
    want = E.compress(y, z, sy, ny, sz, tail=10)
    got = entropy.custom_compress(m, x.cuda(), tail=10)
    for b in range(2):
        assert got["strings"][b][0] == want["strings"][b][0]
        assert got["strings"][b][1] == want["strings"][b][1]
        assert np.array_equal(E.decode_y(got, b, sy[b], ny[b]), y[b])
    assert torch.equal(entropy.custom_decompress(m, got), out["x_hat"].clamp(0, 1))


def test_non_finite_latents_are_reported():
    """A NaN / infinite latent has no integer support (the reference would raise inside int(floor(.)),
    eval_selfcontained_entropy.py:39): the support scan reports width 0 and every later stage flags it."""
    from dsic_amd import entropy
    for bad in (float("nan"), float("inf"), -float("inf")):
        y = torch.zeros((2, 4, 2, 2), device="cuda"); y[1, 2, 0, 1] = bad
        z = torch.zeros((2, 2, 1, 1), device="cuda")
        c = entropy.compress_latents(y, z, torch.ones((2, 4), device="cuda"), torch.full((2, 4), 3.0, device="cuda"),
                                     torch.ones(2, device="cuda"), tail=10, Lmax=64)
        m = c["meta"].cpu().numpy()
        assert m[1, 1] == 0 and m[0, 1] == 21
        with pytest.raises(entropy.EntropyError):
            entropy._check_err(c["err"], "test")
