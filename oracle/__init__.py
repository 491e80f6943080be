"""ctypes loader for the CPU oracle (oracle/vq_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from vq_amd/.  It is the checker, not the product.

Every wrapper cites the reference function it restates (paths under the reference's src/).
Parity status: see the header of vq_oracle.c ("parity unpinned" at the ndarray boundary;
pinned at label level by tests/golden/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvqoracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no GPU involved)."""
    src = os.path.join(_HERE, "vq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "libvqoracle.so"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, u32p, sz = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_size_t
        L.vqo_euclidean_distance.restype = C.c_double
        L.vqo_euclidean_distance.argtypes = [dp, dp, sz]
        L.vqo_generalized_distance.restype = C.c_double
        L.vqo_generalized_distance.argtypes = [dp, sz, sz, dp, dp, dp]
        L.vqo_find_closest_prototype.restype = sz
        L.vqo_find_closest_prototype.argtypes = [dp, sz, sz, dp, dp, sz, dp]
        L.vqo_find_closest_prototype_matched.restype = sz
        L.vqo_find_closest_prototype_matched.argtypes = [dp, u32p, sz, sz, dp, C.c_uint32, C.c_int, dp, sz, dp]
        L.vqo_find_closest_local_prototype.restype = sz
        L.vqo_find_closest_local_prototype.argtypes = [dp, sz, sz, dp, dp, sz, dp]
        L.vqo_find_closest_local_prototype_matched.restype = sz
        L.vqo_find_closest_local_prototype_matched.argtypes = [dp, u32p, sz, sz, dp, C.c_uint32, C.c_int, dp, sz, dp]
        L.vqo_assign_batch.restype = C.c_int
        L.vqo_assign_batch.argtypes = [C.c_void_p, C.c_int, sz, sz, dp, sz, u32p, dp, C.c_int, sz, C.c_int,
                                       u32p, C.c_int, C.c_int, u32p, dp, dp]
        L.vqo_all_distances.restype = C.c_int
        L.vqo_all_distances.argtypes = [dp, sz, dp, sz, dp, C.c_int, sz, dp]
        L.vqo_normalize_omega.restype = None
        L.vqo_normalize_omega.argtypes = [dp, sz, sz]
        L.vqo_project_lambda.restype = C.c_int
        L.vqo_project_lambda.argtypes = [dp, sz, sz, dp, sz, dp]
        L.vqo_project_omega.restype = None
        L.vqo_project_omega.argtypes = [dp, sz, sz, dp, sz, dp]
        L.vqo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def euclidean_distance(v1, v2) -> float:
    """helpers.rs:11-24 (rooted)."""
    v1, v2 = _f64(v1), _f64(v2)
    assert v1.shape == v2.shape and v1.ndim == 1            # helpers.rs:14
    return lib().vqo_euclidean_distance(_d(v1), _d(v2), v1.size)


def generalized_distance(omega, v1, v2) -> float:
    """helpers.rs:35-45 (squared, Lambda rebuilt in the call)."""
    omega, v1, v2 = _f64(omega), _f64(v1), _f64(v2)
    m, n = omega.shape
    scratch = np.empty(n * n + n)
    return lib().vqo_generalized_distance(_d(omega), m, n, _d(v1), _d(v2), _d(scratch))


def normalize_omega(omega):
    """gmlvq.rs:112-122."""
    o = _f64(omega).copy()
    lib().vqo_normalize_omega(_d(o), o.shape[0], o.shape[1])
    return o


def _omega_kind(omega, K):
    if omega is None:
        return None, 0, 0
    om = _f64(omega)
    if om.ndim == 2:
        return om, 1, om.shape[0]
    assert om.ndim == 3 and om.shape[0] == K, "local omegas must be K x M x D"
    return om, 2, om.shape[1]


def find_closest(prototypes, sample, omega=None, proto_cls=None, label=None, matching=True) -> int:
    """One sample through helpers.rs:55-77 / :91-132 or lgmlvq.rs:436-455 / :468-506
    (omega.ndim == 3 selects the local twins; label given selects the matched ones)."""
    W, x = _f64(prototypes), _f64(sample)
    K, D = W.shape
    om, kind, M = _omega_kind(omega, K)
    scratch = np.empty(D * D + D)
    L = lib()
    if label is None:
        f = L.vqo_find_closest_local_prototype if kind == 2 else L.vqo_find_closest_prototype
        return int(f(_d(W), K, D, _d(x), _d(om), M, _d(scratch)))
    cls = np.ascontiguousarray(proto_cls, dtype=np.uint32)
    f = L.vqo_find_closest_local_prototype_matched if kind == 2 else L.vqo_find_closest_prototype_matched
    return int(f(_d(W), _u(cls), K, D, _d(x), int(label), int(bool(matching)), _d(om), M, _d(scratch)))


def assign(X, prototypes, omega=None, proto_cls=None, labels=None, matching=True, hoist=True, threads=1):
    """Batch argmin = the per-sample loop of the six predicts (e.g. glvq.rs:224-232).

    X may be float32 (widened to f64 inside, so device and oracle see identical inputs) or
    float64.  Returns (idx uint32[N], best f64[N], second f64[N]); best/second are the
    winner's and the runner-up's distance among the eligible prototypes (harness extras).
    hoist=False rebuilds Lambda for every (sample, prototype) pair like helpers.rs:38.
    """
    X = np.ascontiguousarray(X)
    if X.dtype != np.float32:
        X = _f64(X)
    W = _f64(prototypes)
    N, D = X.shape
    K = W.shape[0]
    assert W.shape[1] == D
    om, kind, M = _omega_kind(omega, K)
    cls = np.ascontiguousarray(proto_cls, dtype=np.uint32) if proto_cls is not None else None
    lab = np.ascontiguousarray(labels, dtype=np.uint32) if labels is not None else None
    if lab is not None:
        assert cls is not None and lab.shape == (N,)
    idx = np.empty(N, dtype=np.uint32)
    best = np.empty(N)
    second = np.empty(N)
    rc = lib().vqo_assign_batch(X.ctypes.data_as(C.c_void_p), int(X.dtype == np.float32), N, D, _d(W), K,
                                _u(cls), _d(om), kind, M, int(bool(hoist)), _u(lab), int(bool(matching)),
                                int(threads), _u(idx), _d(best), _d(second))
    if rc != 0:
        raise MemoryError("vqo_assign_batch failed")
    return idx, best, second


def all_distances(sample, prototypes, omega=None):
    W, x = _f64(prototypes), _f64(sample)
    K, D = W.shape
    om, kind, M = _omega_kind(omega, K)
    out = np.empty(K)
    if lib().vqo_all_distances(_d(x), D, _d(W), K, _d(om), kind, M, _d(out)) != 0:
        raise MemoryError
    return out


def project_lambda(omega, X):
    """gmlvq.rs:354-372: Lambda . x for every sample."""
    om, X = _f64(omega), _f64(X)
    out = np.empty_like(X)
    if lib().vqo_project_lambda(_d(om), om.shape[0], om.shape[1], _d(X), X.shape[0], _d(out)) != 0:
        raise MemoryError
    return out


def project_omega(omega, X):
    """liramlvq.rs:429-446: Omega . x for every sample."""
    om, X = _f64(omega), _f64(X)
    out = np.empty((X.shape[0], om.shape[0]))
    lib().vqo_project_omega(_d(om), om.shape[0], om.shape[1], _d(X), X.shape[0], _d(out))
    return out


def max_threads() -> int:
    return int(lib().vqo_max_threads())
