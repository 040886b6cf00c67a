"""ctypes loader for the plain-C oracle (oracle/fastmax_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfastmax_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "fastmax_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp, dp, i, d = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                        ctypes.c_double)
        L.fastmax_oracle_fwd.argtypes = [fp, fp, fp, dp, dp, i, i, i, i, i, i, i, d, d, d, i]
        L.fastmax_oracle_bwd.argtypes = [fp, fp, fp, dp, dp, fp, dp, dp, dp, i, i, i, i, i, i, i, d, d, i]
        L.fastmax_oracle_normalize.argtypes = [fp, fp, i, i, i]
        for f in (L.fastmax_oracle_fwd, L.fastmax_oracle_bwd, L.fastmax_oracle_normalize,
                  L.fastmax_oracle_max_threads):
            f.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def max_threads():
    return lib().fastmax_oracle_max_threads()


def fwd(q, k, v, mask=True, nt=None, p=1, g_const=None, nthreads=0):
    """-> (o, g) float64.  Inputs are rounded to float32 first (what the kernels see)."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    nt = 8.0 * np.sqrt(D) if nt is None else float(nt)
    o = np.empty((B, H, Nq, D), dtype=np.float64)
    g = np.empty((B, H, Nq), dtype=np.float64)
    g0 = float(Nq if g_const is None else g_const)
    rc = lib().fastmax_oracle_fwd(_p(q, ctypes.c_float), _p(k, ctypes.c_float), _p(v, ctypes.c_float),
                                  _p(o, ctypes.c_double), _p(g, ctypes.c_double), B, H, Nq, Nk, D, int(p),
                                  int(bool(mask)), 1.0 / nt, 1.0 / (2.0 * nt * nt), g0, int(nthreads))
    if rc != 0:
        raise ValueError(f"fastmax_oracle_fwd rc={rc} (p should be 1 or 2, got p={p})")
    return o, g


def bwd(q, k, v, grad_o, mask=True, nt=None, p=1, g_const=None, nthreads=0):
    """-> (dq, dk, dv) float64."""
    q, k, v, G = _f32(q), _f32(k), _f32(v), _f32(grad_o)
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    nt = 8.0 * np.sqrt(D) if nt is None else float(nt)
    o, g = fwd(q, k, v, mask=mask, nt=nt, p=p, g_const=g_const, nthreads=nthreads)
    dq = np.empty((B, H, Nq, D), dtype=np.float64)
    dk = np.empty((B, H, Nk, D), dtype=np.float64)
    dv = np.empty((B, H, Nk, D), dtype=np.float64)
    c = ctypes
    rc = lib().fastmax_oracle_bwd(_p(q, c.c_float), _p(k, c.c_float), _p(v, c.c_float), _p(o, c.c_double),
                                  _p(g, c.c_double), _p(G, c.c_float), _p(dq, c.c_double), _p(dk, c.c_double),
                                  _p(dv, c.c_double), B, H, Nq, Nk, D, int(p), int(bool(mask)), 1.0 / nt,
                                  1.0 / (2.0 * nt * nt), int(nthreads))
    if rc != 0:
        raise ValueError(f"fastmax_oracle_bwd rc={rc}")
    return dq, dk, dv


def normalize(x):
    x = _f32(x)
    B, H, N, D = x.shape
    y = np.empty_like(x)
    lib().fastmax_oracle_normalize(_p(x, ctypes.c_float), _p(y, ctypes.c_float), B * H, N, D)
    return y
