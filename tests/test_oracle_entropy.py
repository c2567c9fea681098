"""The entropy oracle (oracle/entropy_ref.c) against independent math (scipy) and
an independent pure-Python big-integer range coder.  CPU only."""
import math

import numpy as np
import pytest
from scipy import special

from oracle import entropy_ref as E


def test_elementary_functions_vs_libm():
    L = E.lib()
    for x in [-700.0, -37.5, -1.0, -1e-9, 0.0, 1e-12, 0.3, 1.0, 5.5, 88.0, 700.0]:
        assert abs(L.ora_exp(x) - math.exp(x)) <= 4e-16 * math.exp(x), x
    for x in [1e-300, 1e-5, 0.5, 0.9999999, 1.0, 1.0000001, 1.5, 2.0, 10.0, 12345.678, 1e300]:
        assert abs(L.ora_log(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x))), x
    for x in [0.5, 1.0, 1.5, 2.0, 2.5, 3.7, 10.0, 25.25, 50.0, 50.5, 100.0]:
        assert abs(L.ora_lgamma(x) - math.lgamma(x)) <= 2e-14 * max(1.0, abs(math.lgamma(x))), x


def test_cdfs_vs_scipy():
    xs = np.concatenate([np.linspace(-12, 12, 481), [-40.0, -8.3, 8.3, 40.0]])
    got = E.normal_cdf(xs)
    want = special.ndtr(xs)
    assert np.max(np.abs(got - want)) < 5e-16
    small = (xs < -3) & (want > 0)
    assert np.max(np.abs(got[small] / want[small] - 1.0)) < 1e-11     # left tail keeps relative accuracy
    for nu in [2.0, 2.5, 3.9, 4.0, 7.77, 30.0, 100.0]:
        ts = np.concatenate([np.linspace(-30, 30, 241), [-1e3, -1e-3, 0.0, 1e-3, 1e3]])
        got = E.student_t_cdf(ts, nu)
        want = special.stdtr(nu, ts)
        assert np.max(np.abs(got - want)) < 2e-14, nu
    assert E.student_t_cdf(0.0, 3.3) == 0.5


def _py_table(F, L):
    p = [max(F[k + 1] - F[k], 1e-12) for k in range(L)]
    tot = 0.0
    for v in p:
        tot += v
    cum, out = 0.0, []
    for k in range(L + 1):
        v = cum
        if k == L and v < 1.0:
            v = 1.0
        u16 = int(min(max(v * 65535.0, 0.0), 65535.0))
        out.append(u16 * (65536 - L) // 65535 + k)
        if k < L:
            cum += p[k] / tot
    return out


def test_tables_follow_the_frozen_definition():
    sig = np.array([0.05, 0.4, 1.0, 2.7, 30.0], dtype=np.float32)
    nu = np.array([2.0, 3.3, 4.7, 20.0, 100.0], dtype=np.float32)
    smin, L = -17, 35
    tg = E.tables_gauss(sig, smin, L)
    ts = E.tables_student(sig, nu, smin, L)
    for c in range(sig.size):
        bounds = (np.arange(L + 1) + smin - 0.5) / float(sig[c])
        Fg = [float(v) for v in E.normal_cdf(bounds)]
        Fs = [float(v) for v in E.student_t_cdf(bounds, float(nu[c]))]
        for tab, F in ((tg, Fg), (ts, Fs)):
            want = _py_table(F, L)
            assert want[0] == 0 and want[L] == 65536
            assert list(tab[c]) == want[:L]
            full = np.array(want)
            assert np.all(np.diff(full) >= 1)           # every symbol keeps a non-empty interval
    # against scipy-built tables: same integers except where a cdf value sits within 1e-11 of a
    # truncation boundary (none for these parameters)
    for c in range(sig.size):
        bounds = (np.arange(L + 1) + smin - 0.5) / float(sig[c])
        want = _py_table(list(special.stdtr(float(nu[c]), bounds)), L)
        assert list(ts[c]) == want[:L]


class PyCoder:
    """Independent arithmetic coder on Python integers (no 32-bit wrap tricks)."""

    def __init__(self):
        self.bits = []

    def encode(self, syms, cum):            # cum(i, s) -> (c_low, c_high) in [0, 65536]
        low, high, pending = 0, (1 << 32) - 1, 0
        for i, s in enumerate(syms):
            cl, ch = cum(i, s)
            span = high - low + 1
            high = low + (span * ch >> 16) - 1
            low = low + (span * cl >> 16)
            while True:
                if high < (1 << 31):
                    self._emit(0, pending); pending = 0
                elif low >= (1 << 31):
                    self._emit(1, pending); pending = 0
                    low -= 1 << 31; high -= 1 << 31
                elif low >= (1 << 30) and high < (3 << 30):
                    pending += 1
                    low -= 1 << 30; high -= 1 << 30
                else:
                    break
                low, high = low * 2, high * 2 + 1
        pending += 1
        self._emit(0 if low < (1 << 30) else 1, pending)
        while len(self.bits) % 8:
            self.bits.append(0)
        by = bytearray()
        for k in range(0, len(self.bits), 8):
            v = 0
            for bit in self.bits[k:k + 8]:
                v = v * 2 + bit
            by.append(v)
        return bytes(by)

    def _emit(self, bit, pending):
        self.bits.append(bit)
        self.bits.extend([1 - bit] * pending)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_range_coder_matches_python_integers_and_round_trips(seed):
    rng = np.random.default_rng(seed)
    C, hw, L = 6, 40, 23
    sig = rng.uniform(0.3, 4.0, C).astype(np.float32)
    nu = rng.uniform(2.0, 30.0, C).astype(np.float32)
    tab = E.tables_student(sig, nu, -11, L)
    sym = np.clip(np.rint(rng.standard_t(3.0, size=C * hw) * 1.5), -11, 11).astype(np.int32) + 11
    if seed == 2:
        sym[:] = 11                       # one repeated most-probable symbol: long pending runs stay correct
        sym[::17] = 0                     # and the rarest one
    data = E.range_encode(sym, tab, hw)

    def cum(i, s):
        t = tab[i // hw]
        return int(t[s]), (65536 if s == L - 1 else int(t[s + 1]))

    assert data == PyCoder().encode(list(sym), cum)
    back = E.range_decode(data, sym.size, tab, hw)
    assert np.array_equal(back, sym)
    bits = -np.log2(np.array([(cum(i, s)[1] - cum(i, s)[0]) / 65536 for i, s in enumerate(sym)])).sum()
    assert len(data) * 8 <= bits + 16 and len(data) * 8 >= bits - 8


def test_edge_cases():
    tab = E.tables_gauss(np.array([1.0], dtype=np.float32), -10, 21)
    one = E.range_encode(np.array([10], dtype=np.int32), tab, 1)
    assert E.range_decode(one, 1, tab, 1)[0] == 10
    # sigma tiny: all mass in one bin, the others live on the spread floor
    t2 = E.tables_gauss(np.array([1e-3], dtype=np.float32), -10, 21)
    d = np.diff(np.append(t2[0].astype(np.int64), 65536))
    assert d.min() >= 1 and d.max() > 65000
    sym = np.array([10] * 50 + [0, 20, 10], dtype=np.int32)
    assert np.array_equal(E.range_decode(E.range_encode(sym, t2, sym.size), sym.size, t2, sym.size), sym)


def test_compress_flow_round_trip():
    rng = np.random.default_rng(5)
    B, M, N = 2, 8, 4
    y = np.rint(rng.standard_t(4.0, size=(B, M, 4, 6)) * 2).astype(np.float32)
    z = np.rint(rng.normal(size=(B, N, 1, 2)) * 3).astype(np.float32)
    sy = rng.uniform(1.0, 3.0, (B, M)).astype(np.float32)
    ny = rng.uniform(2.0, 9.0, (B, M)).astype(np.float32)
    sz = rng.uniform(1.0, 4.0, N).astype(np.float32)
    c = E.compress(y, z, sy, ny, sz, tail=10)
    assert c["shape_y"] == [B, M, 4, 6] and len(c["strings"]) == B
    for b in range(B):
        assert c["min_y"][b] == int(y[b].min()) - 10 and c["max_z"][b] == int(z[b].max()) + 10
        assert np.array_equal(E.decode_z(c, b, sz), z[b])
        assert np.array_equal(E.decode_y(c, b, sy[b], ny[b]), y[b])
