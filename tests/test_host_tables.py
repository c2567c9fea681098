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
    assert L.dsic_abi_version() == 1


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
            assert Lh.dsic_host_cdf_table(student, float(sig), float(nu), smin, Lsym,
                                          out.ctypes.data_as(ctypes.c_void_p)) == 0
            want = (E.tables_student(np.array([sig]), np.array([nu]), smin, Lsym) if student
                    else E.tables_gauss(np.array([sig]), smin, Lsym))[0]
            assert np.array_equal(out, want)
