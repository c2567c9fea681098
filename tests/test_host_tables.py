"""CPU-only: the C-ABI library loads, exports every declared symbol, and its
host-side table math equals the oracle bit for bit (same frozen algorithm,
independent implementation)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import entropy_ref as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from dsic_amd import lib
    return lib.load()


def test_library_exports_every_declared_symbol(L):
    from dsic_amd import lib
    header = open(os.path.join(ROOT, "include", "dsic_hip.h")).read()
    declared = set(re.findall(r"\b(dsic_[a-zA-Z0-9_]+)\s*\(", header))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.dsic_abi_version() == 4


def test_argument_validation_without_gpu(L):
    # NULL pointers / bad shapes are rejected before anything touches the device
    assert L.dsic_conv2d_nhwc(None, None, None, None, None, None, 1, 8, 8, 8, 8, 3, 1, 0, None) == 1
    assert b"null" in L.dsic_last_error()
    assert L.dsic_range_encode(None, None, None, None, None, 64, 1, 1, 1, 1, 1, None, 8, 8, None, None, 4, 0, None) == 1
    assert L.dsic_packed_conv_weight_floats(128, 3, 3) == 9 * 1 * 128 * 8


def test_host_cdfs_equal_oracle_bits(L):
    xs = np.concatenate([np.linspace(-9, 9, 181), [-38.0, 0.0, 1e-9, 38.0]])
    for x in xs:
        assert L.dsic_host_normal_cdf(float(x)) == E.lib().ora_normal_cdf(float(x))
    for nu in (2.0, 2.7, 4.0, 11.5, 100.0):
        for t in np.linspace(-25, 25, 101):
            assert L.dsic_host_student_t_cdf(float(t), nu) == E.lib().ora_student_t_cdf(float(t), nu)


def test_host_tables_equal_oracle():
    from dsic_amd import lib
    Lh = lib.load()
    rng = np.random.default_rng(0)
    for _ in range(20):
        sig = np.float32(np.exp(rng.uniform(-3, 3)))
        nu = np.float32(rng.uniform(2, 100))
        smin = int(rng.integers(-40, -5))
        Lsym = int(rng.integers(21, 90))
        for student in (0, 1):
            out = np.zeros(Lsym, dtype=np.uint16)
            raw = np.zeros(Lsym + 1, dtype=np.uint16)
            assert Lh.dsic_host_cdf_table(student, float(sig), float(nu), smin, Lsym,
                                          out.ctypes.data_as(ctypes.c_void_p),
                                          raw.ctypes.data_as(ctypes.c_void_p)) == 0
            want, want_raw = (E.tables_student(np.array([sig]), np.array([nu]), smin, Lsym, raw=True) if student
                              else E.tables_gauss(np.array([sig]), smin, Lsym, raw=True))
            assert np.array_equal(out, want[0]) and np.array_equal(raw, want_raw[0])


# ---- the product's host table math against the REFERENCE-generated fixture ------------------
GOLD = os.path.join(ROOT, "tests", "golden", "entropy_ref.npz")


def test_product_pmf_to_uint16_cdf_reproduces_the_reference():
    """eval_selfcontained_entropy.py:17-23, outputs of the reference's own function."""
    from dsic_amd import entropy
    g = np.load(GOLD)
    for i in range(int(g["u16cdf/count"][0])):
        got = entropy.pmf_to_uint16_cdf(g[f"u16cdf/{i}/pmf"])
        assert got.dtype == np.uint16 and np.array_equal(got, g[f"u16cdf/{i}/out"]), i


def test_product_gaussian_cdf_vs_reference():
    from dsic_amd import entropy
    g = np.load(GOLD)
    x, want = g["gcdf/x"], g["gcdf/y"]
    got = entropy.gaussian_cdf(x)
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), E.gaussian_cdf_f32(x).view(np.uint32))    # host == oracle bits
    assert np.max(np.abs(got.astype(np.float64) - want)) <= 2.0 ** -24                 # erf's last bit
    assert np.count_nonzero(got != want) <= 0.03 * x.size


def test_product_z_tables_vs_reference(L):
    """Pre-spreading uint16 tables of the reference's z flow (:36-47): max |diff| 1 on < 0.1 % of
    the entries (erf's last bit, see tests/test_oracle_entropy.py), identical to the oracle."""
    g = np.load(GOLD)
    total = bad = 0
    for i in range(int(g["zmodel/count"][0])):
        sig, ref = g[f"zmodel/{i}/sigma_z"], g[f"zmodel/{i}/cdf_u16"]
        zmin, Ls = int(g[f"zmodel/{i}/zmin"][0]), ref.shape[0] - 1
        _, oraw = E.tables_gauss(sig, zmin, Ls, raw=True)
        for c in range(sig.size):
            out = np.zeros(Ls, dtype=np.uint16)
            raw = np.zeros(Ls + 1, dtype=np.uint16)
            assert L.dsic_host_cdf_table(0, float(sig[c]), 0.0, zmin, Ls, out.ctypes.data_as(ctypes.c_void_p),
                                         raw.ctypes.data_as(ctypes.c_void_p)) == 0
            assert np.array_equal(raw, oraw[c])
            d = np.abs(raw.astype(np.int64) - ref[:, c].astype(np.int64))
            assert d.max() <= 1
            bad += np.count_nonzero(d)
            total += d.size
    assert bad <= 0.001 * total, (bad, total)


def test_sigma_z_is_deterministic_and_within_one_ulp_of_torch(L):
    """:32 sigma_z = exp(log_sigma): the library's float32(exp64) against the reference's CPU torch
    values (MKL VML vsExp, < 1 ulp)."""
    g = np.load(GOLD)
    n = diff = 0
    for i in range(int(g["zsweep/count"][0])):
        ls, want = g[f"zsweep/{i}/log_sigma"], g[f"zsweep/{i}/sigma_z"]
        got = np.array([L.dsic_host_exp_f32(float(v)) for v in ls], dtype=np.float32)
        assert np.all(np.abs(got.astype(np.float64) - want) <= np.spacing(want))
        assert np.array_equal(got, np.exp(ls.astype(np.float64)).astype(np.float32))     # correctly rounded
        diff += np.count_nonzero(got != want)
        n += ls.size
    assert diff <= 0.02 * n, (diff, n)


def test_split_k_policy_depends_on_the_layer_geometry_only(L):
    """dsic_wino_bf16_ksplit(H, W, Cin): layers with fewer than four 16x8-pixel tiles per image share a tile's
    input channels between work items, in even runs of >= 4 sixteen-channel chunks.  There is no batch argument:
    a patch's bits cannot depend on the batch it is in (tests/test_gpu_fullsize.py checks the invariance)."""
    ks = L.dsic_wino_bf16_ksplit
    assert ks(128, 128, 128) == 1 and ks(32, 32, 512) == 1          # many tiles per image
    assert ks(16, 16, 512) == 2                                     # g_a.14 at 256x256: 2 tiles per image
    assert ks(8, 8, 512) == 4 and ks(4, 4, 512) == 4                # h_a.4 / h_a.6: one (partial) tile
    assert ks(16, 16, 128) == 2 and ks(16, 16, 192) == 2            # 8 / 12 chunks -> runs of 4 / 6
    assert ks(8, 8, 128) == 2                                       # 8 chunks cannot make four runs of >= 4
    assert ks(8, 8, 64) == 1 and ks(8, 8, 96) == 1                  # 4 / 6 chunks: no even split of >= 4
    assert ks(0, 8, 128) == 1 and ks(8, 8, 100) == 1                # nonsense stays unsplit
    for H, W, C in ((16, 16, 512), (8, 8, 512), (16, 16, 192), (24, 16, 256), (8, 40, 128)):
        S = ks(H, W, C)
        n = C // 16
        assert S >= 1 and n % S == 0 and (S == 1 or ((n // S) >= 4 and (n // S) % 2 == 0))
