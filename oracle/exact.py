"""ctypes front end of oracle/exact_c/pairwise_exact.c - the plain-C model of the
device kernels' f32 summation order for the PRDC path (bit-exact checker).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exact_c")
_SO = os.path.join(_DIR, "libpairwise_exact.so")
_lib = None


def build(force=False):
    src = os.path.join(_DIR, "pairwise_exact.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _DIR, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _mat(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64


def sqnorm(x):
    x = _mat(x)
    out = np.empty(len(x), dtype=np.float32)
    _load().am_exact_sqnorm(_p(x), _i64(len(x)), _i64(x.shape[1]), ctypes.c_int(x.shape[1]), _p(out))
    return out


def threshold(r):
    r = np.ascontiguousarray(r, dtype=np.float32)
    out = np.empty_like(r)
    _load().am_exact_threshold(_p(r), _i64(len(r)), _p(out))
    return out


def knn_radii(x, k, columns=None, return_squared=False):
    x = _mat(x)
    y = x if columns is None else _mat(columns)
    out = np.empty(len(x), dtype=np.float32)
    out2 = np.empty(len(x), dtype=np.float32)
    rc = _load().am_exact_knn_radii(_p(x), _i64(len(x)), _i64(x.shape[1]), _p(y), _i64(len(y)), _i64(y.shape[1]),
                                    ctypes.c_int(x.shape[1]), ctypes.c_int(k), _p(out), _p(out2))
    if rc != 0:
        raise ValueError(f"k + 1 = {k + 1} exceeds the {len(y)} available rows")
    return (out, out2) if return_squared else out


def prdc_counts(ref, cand, r_ref, r_cand):
    ref, cand = _mat(ref), _mat(cand)
    r_ref = np.ascontiguousarray(r_ref, dtype=np.float32)
    r_cand = np.ascontiguousarray(r_cand, dtype=np.float32)
    col = np.empty(len(cand), dtype=np.int32)
    rany = np.empty(len(ref), dtype=np.uint8)
    rmin = np.empty(len(ref), dtype=np.float32)
    _load().am_exact_prdc_counts(_p(ref), _i64(len(ref)), _i64(ref.shape[1]), _p(cand), _i64(len(cand)),
                                 _i64(cand.shape[1]), ctypes.c_int(ref.shape[1]), _p(r_ref), _p(r_cand),
                                 _p(col), _p(rany), _p(rmin))
    return col, rany, rmin


def prdc(ref, cand, k):
    """Full PRDC through the device-order model; final means as prdc.py:36-48."""
    r_ref, r_cand = knn_radii(ref, k), knn_radii(cand, k)
    col, rany, rmin = prdc_counts(ref, cand, r_ref, r_cand)
    return dict(precision=float((col > 0).astype(np.float64).mean()),
                recall=float(rany.astype(np.float64).mean()),
                density=(1.0 / float(k)) * float(col.astype(np.float64).mean()),
                coverage=float((rmin < r_ref).astype(np.float64).mean())), dict(
                    r_ref=r_ref, r_cand=r_cand, col_count=col, row_any=rany, row_min=rmin)
