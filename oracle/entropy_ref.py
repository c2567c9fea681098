"""ORACLE (test infrastructure): ctypes front end of oracle/entropy_ref.c plus the
per-image compress/decompress flow of eval_selfcontained_entropy.py:26-123
restated on numpy arrays.  See the header of entropy_ref.c for the parity
status (z tables pinned by tests/golden/entropy_ref.npz; Student-t tables and
coded bytes unpinned: that part of the reference cannot execute) and the
frozen interpretation choices."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import build_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build_oracle.LIB
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(build_oracle.SRC):
            path = build_oracle.build()
        L = ctypes.CDLL(path)
        d = ctypes.c_double
        for name in ("ora_exp", "ora_log", "ora_lgamma", "ora_normal_cdf"):
            getattr(L, name).restype = d
            getattr(L, name).argtypes = [d]
        L.ora_student_t_cdf.restype = d
        L.ora_student_t_cdf.argtypes = [d, d]
        vp, i, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.ora_tables_gauss.restype = None
        L.ora_tables_gauss.argtypes = [vp, i, i, i, vp, vp, vp]
        L.ora_tables_student.restype = None
        L.ora_tables_student.argtypes = [vp, vp, i, i, i, vp, vp, vp]
        f = ctypes.c_float
        L.ora_erff.restype = f
        L.ora_erff.argtypes = [f]
        L.ora_gaussian_cdf_f32.restype = f
        L.ora_gaussian_cdf_f32.argtypes = [f]
        L.ora_sum_f32.restype = f
        L.ora_sum_f32.argtypes = [vp, i]
        L.ora_pmf_to_uint16_cdf.restype = None
        L.ora_pmf_to_uint16_cdf.argtypes = [vp, i, vp]
        L.ora_range_encode.restype = i64
        L.ora_range_encode.argtypes = [vp, i64, vp, i, i, vp, i64]
        L.ora_range_decode.restype = None
        L.ora_range_decode.argtypes = [vp, i64, i64, vp, i, i, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def normal_cdf(x):
    L = lib()
    return np.array([L.ora_normal_cdf(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def student_t_cdf(t, nu):
    L = lib()
    t, nu = np.broadcast_arrays(np.asarray(t, dtype=np.float64), np.asarray(nu, dtype=np.float64))
    return np.array([L.ora_student_t_cdf(float(a), float(b)) for a, b in zip(t.ravel(), nu.ravel())]).reshape(t.shape)


def gaussian_cdf_f32(x):
    """eval_selfcontained_entropy.py:14-15 evaluated in float32 (x float32 array)."""
    L = lib()
    x = np.asarray(x, dtype=np.float32)
    return np.array([L.ora_gaussian_cdf_f32(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def sum_f32(p):
    """torch.sum(dim=0) order for one float32 column."""
    p = np.ascontiguousarray(p, dtype=np.float32)
    return np.float32(lib().ora_sum_f32(_ptr(p), p.size))


def pmf_to_uint16_cdf(pmf):
    """:17-23; pmf [L, C] float32 -> uint16 [L+1, C]."""
    pmf = np.asarray(pmf, dtype=np.float32)
    L, C = pmf.shape
    out = np.empty((C, L + 1), dtype=np.uint16)
    cols = np.ascontiguousarray(pmf.T)
    for c in range(C):
        lib().ora_pmf_to_uint16_cdf(_ptr(cols[c]), L, _ptr(out[c]))
    return np.ascontiguousarray(out.T)


def tables_gauss(sigma, smin, L, raw=False):
    """coder tables [C, L]; raw=True also returns the pre-spreading uint16 cdf [C, L+1] of :22."""
    sigma = np.ascontiguousarray(sigma, dtype=np.float32)
    C = sigma.size
    out = np.empty((C, L), dtype=np.uint16)
    r = np.empty((C, L + 1), dtype=np.uint16) if raw else None
    work = np.empty(3 * L + 4, dtype=np.float32)
    lib().ora_tables_gauss(_ptr(sigma), C, int(smin), int(L), _ptr(out), _ptr(r) if raw else None, _ptr(work))
    return (out, r) if raw else out


def tables_student(sigma, nu, smin, L, raw=False):
    sigma = np.ascontiguousarray(sigma, dtype=np.float32)
    nu = np.ascontiguousarray(nu, dtype=np.float32)
    C = sigma.size
    out = np.empty((C, L), dtype=np.uint16)
    r = np.empty((C, L + 1), dtype=np.uint16) if raw else None
    work = np.empty(3 * L + 4, dtype=np.float32)
    lib().ora_tables_student(_ptr(sigma), _ptr(nu), C, int(smin), int(L), _ptr(out),
                             _ptr(r) if raw else None, _ptr(work))
    return (out, r) if raw else out


def range_encode(sym, tables, hw):
    sym = np.ascontiguousarray(sym, dtype=np.int32).ravel()
    tables = np.ascontiguousarray(tables, dtype=np.uint16)
    Lsym = tables.shape[1]
    assert sym.min() >= 0 and sym.max() < Lsym
    cap = 2 * sym.size + 16
    out = np.empty(cap, dtype=np.uint8)
    n = lib().ora_range_encode(_ptr(sym), sym.size, _ptr(tables), Lsym, int(hw), _ptr(out), cap)
    assert n >= 0
    return out[:n].tobytes()


def range_decode(data, n, tables, hw):
    buf = np.frombuffer(data, dtype=np.uint8)
    tables = np.ascontiguousarray(tables, dtype=np.uint16)
    sym = np.empty(n, dtype=np.int32)
    lib().ora_range_decode(_ptr(buf), buf.size, int(n), _ptr(tables), tables.shape[1], int(hw), _ptr(sym))
    return sym


def support(values, tail):
    """eval_selfcontained_entropy.py:39-41 / 52-54."""
    vmin = int(np.floor(values.min())) - tail
    vmax = int(np.ceil(values.max())) + tail
    return vmin, vmax


def compress(y_q, z_q, sigma_y, nu_y, sigma_z, tail=10):
    """custom_compress (:26-74) after the forward pass: y_q [B,M,Hy,Wy], z_q [B,N,Hz,Wz]
    integer-valued float arrays, sigma_y/nu_y [B,M] float32, sigma_z [N] float32."""
    B = y_q.shape[0]
    strings, miny, maxy, minz, maxz = [], [], [], [], []
    for b in range(B):
        zmin, zmax = support(z_q[b], tail)
        tz = tables_gauss(sigma_z, zmin, zmax - zmin + 1)
        zs = range_encode(z_q[b].astype(np.int32) - zmin, tz, z_q.shape[2] * z_q.shape[3])
        ymin, ymax = support(y_q[b], tail)
        if np.ndim(sigma_y[b]) == 3:   # spatial_params: StudentT(df=nu_y[b], scale=sigma_y[b]) per element (:57)
            ty = tables_student(np.ravel(sigma_y[b]), np.ravel(nu_y[b]), ymin, ymax - ymin + 1)
            ys = range_encode(y_q[b].astype(np.int32) - ymin, ty, 1)
        else:
            ty = tables_student(sigma_y[b], nu_y[b], ymin, ymax - ymin + 1)
            ys = range_encode(y_q[b].astype(np.int32) - ymin, ty, y_q.shape[2] * y_q.shape[3])
        strings.append([zs, ys])
        miny.append(ymin); maxy.append(ymax); minz.append(zmin); maxz.append(zmax)
    return {"strings": strings, "shape_y": list(y_q.shape), "shape_z": list(z_q.shape),
            "min_y": miny, "max_y": maxy, "min_z": minz, "max_z": maxz}


def decode_z(compressed, b, sigma_z):
    """:88-97: z symbols of image b -> z_hat [N,Hz,Wz] float32."""
    _, N, Hz, Wz = compressed["shape_z"]
    zmin, zmax = compressed["min_z"][b], compressed["max_z"][b]
    tz = tables_gauss(sigma_z, zmin, zmax - zmin + 1)
    s = range_decode(compressed["strings"][b][0], N * Hz * Wz, tz, Hz * Wz)
    return (s + zmin).astype(np.float32).reshape(N, Hz, Wz)


def decode_y(compressed, b, sigma_y, nu_y):
    """:108-117: y symbols of image b given its sigma/nu [M]."""
    _, M, Hy, Wy = compressed["shape_y"]
    ymin, ymax = compressed["min_y"][b], compressed["max_y"][b]
    per_element = np.ndim(sigma_y) == 3
    ty = tables_student(np.ravel(sigma_y), np.ravel(nu_y), ymin, ymax - ymin + 1)
    s = range_decode(compressed["strings"][b][1], M * Hy * Wy, ty, 1 if per_element else Hy * Wy)
    return (s + ymin).astype(np.float32).reshape(M, Hy, Wy)
