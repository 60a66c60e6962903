"""ctypes wrapper of oracle/normal_eq.c (oracle; tests and bench cpu_baseline only)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle_c.so")
_dll = None


def _load():
    global _dll
    if _dll is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} missing: run `make -C oracle`")
        _dll = C.CDLL(_PATH)
        _dll.oracle_normal_eq.restype = C.c_int
        _dll.oracle_normal_eq.argtypes = [C.c_int, C.c_int64] + [C.c_void_p] * 5 + [C.c_double] + [C.c_void_p] * 8 + [C.c_int]
    return _dll


def normal_eq(model, view_ptr, obj, img, intr, poses, lam=0.0, nthreads=1, want_blocks=True):
    """One normal-equation evaluation + pose-block elimination in C.
    model 0 pinhole / 1 fisheye.  Returns dict U,b,cost,C,d (and V,W,g)."""
    dll = _load()
    NI = 9 if model == 0 else 8
    view_ptr = np.ascontiguousarray(view_ptr, dtype=np.int64)
    obj = np.ascontiguousarray(obj, dtype=np.float32).reshape(-1, 3)
    img = np.ascontiguousarray(img, dtype=np.float32).reshape(-1, 2)
    intr = np.ascontiguousarray(intr, dtype=np.float64)
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
    N = view_ptr.shape[0] - 1
    U = np.zeros((NI, NI)); b = np.zeros(NI); cost = np.zeros(1); Cm = np.zeros((NI, NI)); d = np.zeros(NI)
    V = np.zeros((N, 6, 6)) if want_blocks else None
    W = np.zeros((N, NI, 6)) if want_blocks else None
    g = np.zeros((N, 6)) if want_blocks else None
    p = lambda a: a.ctypes.data if a is not None else None
    rc = dll.oracle_normal_eq(model, N, p(view_ptr), p(obj), p(img), p(intr), p(poses), float(lam),
                              p(U), p(b), p(cost), p(V), p(W), p(g), p(Cm), p(d), int(nthreads))
    if rc:
        raise RuntimeError(f"oracle_normal_eq failed: {rc}")
    return dict(U=U, b=b, cost=float(cost[0]), C=Cm, d=d, V=V, W=W, g=g)
