"""Thin Python handle over the C ABI (include/vqhip.h).  PyTorch appears only as the owner of
device memory and streams; numpy arrays take the pinned host path of the library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import VqhipError


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class Context:
    """One GPU, one caller at a time (like `predict(&mut self)` in the reference)."""

    def __init__(self, device: int = 0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.vqhip_create(int(device), C.byref(h))
        if rc != 0:
            raise VqhipError(rc, self._L.vqhip_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.K = self.D = self.M = 0

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.vqhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc: int):
        if rc != 0:
            raise VqhipError(rc, self._L.vqhip_last_error(self._h).decode())

    # -- model state
    def set_prototypes(self, W):
        W = np.ascontiguousarray(W, dtype=np.float32)
        if W.ndim != 2:
            raise ValueError("prototypes must be K x D")
        self._chk(self._L.vqhip_set_prototypes(self._h, _fp(W), W.shape[0], W.shape[1]))
        self.K, self.D, self.M = W.shape[0], W.shape[1], 0

    def set_labels(self, cls):
        cls = np.ascontiguousarray(cls, dtype=np.uint32)
        self._chk(self._L.vqhip_set_labels(self._h, _up(cls), cls.shape[0]))

    def set_omega(self, omega):
        if omega is None:
            self._chk(self._L.vqhip_set_omega(self._h, None, 0, 0))
            self.M = 0
            return
        om = np.ascontiguousarray(omega, dtype=np.float32)
        if om.ndim != 2:
            raise ValueError("omega must be M x D")
        self._chk(self._L.vqhip_set_omega(self._h, _fp(om), om.shape[0], om.shape[1]))
        self.M = om.shape[0]

    def set_local_omegas(self, omegas):
        if omegas is None:
            self._chk(self._L.vqhip_set_local_omegas(self._h, None, 0, 0, 0))
            self.M = 0
            return
        om = np.ascontiguousarray(omegas, dtype=np.float32)
        if om.ndim != 3:
            raise ValueError("local omegas must be K x M x D")
        self._chk(self._L.vqhip_set_local_omegas(self._h, _fp(om), om.shape[0], om.shape[1], om.shape[2]))
        self.M = om.shape[1]

    def set_kernel_policy(self, policy: int):
        self._chk(self._L.vqhip_set_kernel_policy(self._h, int(policy)))

    def set_option(self, option: int, value: int):
        """vqhip_set_option: OPT_RANKING / OPT_PROJECTION -> VALUE_DEFAULT / VALUE_F32 / VALUE_BF16 (images are rebuilt)."""
        self._chk(self._L.vqhip_set_option(self._h, int(option), int(value)))

    @property
    def last_kernel(self) -> str:
        return self._L.vqhip_last_kernel(self._h).decode()

    @property
    def last_ranking(self) -> str:
        """Where the last assign call ranked its candidates: "f32", "bf16x3" or "none" (vqhip_last_ranking)."""
        return self._L.vqhip_last_ranking(self._h).decode()

    @property
    def last_projection(self) -> str:
        """Where the last assign call computed y = Omega x: "f32", "bf16x6" or "none" (vqhip_last_projection)."""
        return self._L.vqhip_last_projection(self._h).decode()

    # -- hot path
    def assign(self, X, want_dist: bool = True):
        """find_closest_prototype for every row of X -> (idx, dist).
        numpy in -> numpy out (pinned chunked uploads); CUDA torch tensor in -> torch tensors out,
        enqueued on torch's current stream with no host synchronisation."""
        if _is_torch(X):
            return self._assign_torch(X, None, want_dist)
        X = np.ascontiguousarray(X, dtype=np.float32)
        self._check_shape(X.shape)
        N = X.shape[0]
        idx = np.empty(N, dtype=np.uint32)
        dist = np.empty(N, dtype=np.float32) if want_dist else None
        self._chk(self._L.vqhip_assign(self._h, _fp(X), N, _up(idx), _fp(dist)))
        return idx, dist

    def assign_rows(self, rows, want_dist: bool = True):
        """`predict`'s own input shape: a sequence of separately allocated f64 rows (the reference's
        `&Vec<Array1<f64>>`, src/vq/vq.rs:142).  Every row's length is asserted here, then the library
        gathers and narrows them straight into its pinned upload buffers."""
        rows = [np.ascontiguousarray(r, dtype=np.float64) for r in rows]
        N = len(rows)
        for r in rows:
            if r.ndim != 1 or r.shape[0] != self.D:
                raise ValueError(f"every row must hold {self.D} values")
        ptrs = (C.POINTER(C.c_double) * max(N, 1))(*[r.ctypes.data_as(C.POINTER(C.c_double)) for r in rows])
        idx = np.empty(N, dtype=np.uint32)
        dist = np.empty(N, dtype=np.float32) if want_dist else None
        self._chk(self._L.vqhip_assign_rows_f64(self._h, ptrs, N, self.D, _up(idx), _fp(dist)))
        return idx, dist

    def project(self, X, which: int = _lib.PROJECT_LAMBDA):
        """project (Lambda x, N x D) / project_omega (Omega x, N x M); local Omegas use the winner's Lambda_j."""
        X = np.ascontiguousarray(X, dtype=np.float32)
        self._check_shape(X.shape)
        width = self.M if which == _lib.PROJECT_OMEGA else self.D
        Y = np.empty((X.shape[0], width), dtype=np.float32)
        self._chk(self._L.vqhip_project(self._h, _fp(X), X.shape[0], int(which), _fp(Y)))
        return Y

    def project_rows(self, rows, which: int = _lib.PROJECT_LAMBDA):
        """project / project_omega for the reference's container on both sides: separately allocated f64 rows in, a list of
        separately allocated f64 rows out (src/gmlvq/gmlvq.rs:354-372)."""
        rows = [np.ascontiguousarray(r, dtype=np.float64) for r in rows]
        N = len(rows)
        for r in rows:
            if r.ndim != 1 or r.shape[0] != self.D:
                raise ValueError(f"every row must hold {self.D} values")
        width = self.M if which == _lib.PROJECT_OMEGA else self.D
        out = [np.empty(width, dtype=np.float64) for _ in range(N)]
        ptrs = (C.POINTER(C.c_double) * max(N, 1))(*[r.ctypes.data_as(C.POINTER(C.c_double)) for r in rows])
        optrs = (C.POINTER(C.c_double) * max(N, 1))(*[r.ctypes.data_as(C.POINTER(C.c_double)) for r in out])
        self._chk(self._L.vqhip_project_rows_f64(self._h, ptrs, N, self.D, int(which), optrs))
        return out

    # -- the Euclidean path in f64 (bit-exact against the reference's arithmetic)
    def set_prototypes_f64(self, W):
        W = np.ascontiguousarray(W, dtype=np.float64)
        if W.ndim != 2:
            raise ValueError("prototypes must be K x D")
        self._chk(self._L.vqhip_set_prototypes_f64(self._h, W.ctypes.data_as(C.POINTER(C.c_double)), W.shape[0],
                                                   W.shape[1]))
        self.K, self.D, self.M = W.shape[0], W.shape[1], 0

    def assign_f64(self, X, y=None):
        """-> (idx, dist) or, with class ids y, (jpos, dpos, jneg, dneg); dist in f64."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        self._check_shape(X.shape)
        N = X.shape[0]
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        idx, dist = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.float64)
        if y is None:
            self._chk(self._L.vqhip_assign_f64(self._h, dp(X), N, _up(idx), dp(dist)))
            return idx, dist
        y = np.ascontiguousarray(y, dtype=np.uint32)
        if y.shape != (N,):
            raise ValueError("y must have one class id per sample")
        jn, dn = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.float64)
        self._chk(self._L.vqhip_assign_matched_f64(self._h, dp(X), _up(y), N, _up(idx), dp(dist), _up(jn), dp(dn)))
        return idx, dist, jn, dn

    def assign_matched(self, X, y):
        """Both fit-time searches -> (jpos, dpos, jneg, dneg)."""
        if _is_torch(X):
            return self._assign_torch(X, y, True)
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.uint32)
        self._check_shape(X.shape)
        N = X.shape[0]
        if y.shape != (N,):
            raise ValueError("y must have one class id per sample")
        jp, jn = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint32)
        dp, dn = np.empty(N, dtype=np.float32), np.empty(N, dtype=np.float32)
        self._chk(self._L.vqhip_assign_matched(self._h, _fp(X), _up(y), N, _up(jp), _fp(dp), _up(jn), _fp(dn)))
        return jp, dp, jn, dn

    def assign_matched_rows(self, rows, y):
        """The matched search for the reference's own container (separately allocated f64 rows)."""
        rows = [np.ascontiguousarray(r, dtype=np.float64) for r in rows]
        N = len(rows)
        for r in rows:
            if r.ndim != 1 or r.shape[0] != self.D:
                raise ValueError(f"every row must hold {self.D} values")
        y = np.ascontiguousarray(y, dtype=np.uint32)
        if y.shape != (N,):
            raise ValueError("y must have one class id per sample")
        ptrs = (C.POINTER(C.c_double) * max(N, 1))(*[r.ctypes.data_as(C.POINTER(C.c_double)) for r in rows])
        jp, jn = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint32)
        dp, dn = np.empty(N, dtype=np.float32), np.empty(N, dtype=np.float32)
        self._chk(self._L.vqhip_assign_matched_rows_f64(self._h, ptrs, _up(y), N, self.D, _up(jp), _fp(dp), _up(jn), _fp(dn)))
        return jp, dp, jn, dn

    def glvq_batch_gradient(self, X, y):
        """vqhip_glvq_batch_gradient (batch form of the two GLVQ update terms, prototypes fixed: NOT the reference's per-sample
        SGD) -> (dW K x D, cost, samples used).  numpy in; CUDA tensors in -> (dW tensor, cost_and_count tensor of 2 x 8 bytes)."""
        if _is_torch(X):
            import torch

            self._check_torch(X)
            N = X.shape[0]
            if not (y.is_cuda and y.device == X.device and y.dtype == torch.int32 and y.is_contiguous() and tuple(y.shape) == (N,)):
                raise ValueError("y must be a contiguous CUDA int32 tensor with one class id per sample")
            dW = torch.empty((self.K, self.D), dtype=torch.float32, device=X.device)
            tail = torch.empty(2, dtype=torch.float64, device=X.device)
            stream = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
            self._chk(self._L.vqhip_glvq_batch_gradient_device(self._h, C.c_void_p(X.data_ptr()), C.c_void_p(y.data_ptr()), N,
                                                               C.c_void_p(dW.data_ptr()), C.c_void_p(tail.data_ptr()), stream))
            return dW, tail
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.uint32)
        self._check_shape(X.shape)
        if y.shape != (X.shape[0],):
            raise ValueError("y must have one class id per sample")
        dW = np.empty((self.K, self.D), dtype=np.float32)
        cost, used = C.c_double(0.0), C.c_uint64(0)
        self._chk(self._L.vqhip_glvq_batch_gradient(self._h, _fp(X), _up(y), X.shape[0], _fp(dW), C.byref(cost), C.byref(used)))
        return dW, cost.value, used.value

    def assign_into(self, X, idx_out, dist_out=None, y=None, jneg_out=None, dneg_out=None):
        """Device path with caller-provided torch output tensors (int32 storage for indices)."""
        import torch

        self._check_torch(X)
        N = X.shape[0]
        self._check_outputs(N, idx_out, dist_out, jneg_out, dneg_out)
        if y is not None and not (y.is_cuda and y.device == X.device and y.dtype == torch.int32 and y.is_contiguous()
                                  and tuple(y.shape) == (N,)):
            raise ValueError("y must be a contiguous CUDA int32 tensor with one class id per sample")
        stream = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)

        def p(t):
            return C.c_void_p(t.data_ptr()) if t is not None else None

        if y is None:
            self._chk(self._L.vqhip_assign_device(self._h, p(X), N, p(idx_out), p(dist_out), stream))
        else:
            self._chk(self._L.vqhip_assign_matched_device(self._h, p(X), p(y), N, p(idx_out), p(dist_out),
                                                          p(jneg_out), p(dneg_out), stream))

    def prepare(self):
        """Builds the model's device images now (vqhip_prepare)."""
        self._chk(self._L.vqhip_prepare(self._h))

    def assign_into_noalloc(self, X, idx_out, dist_out=None, n=None):
        """vqhip_assign_device_noalloc: the device path that allocates nothing (the model must be prepared)."""
        import torch

        self._check_torch(X)
        N = X.shape[0] if n is None else int(n)
        self._check_outputs(N if n is None else 0, idx_out, dist_out)
        stream = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
        self._chk(self._L.vqhip_assign_device_noalloc(self._h, C.c_void_p(X.data_ptr()), N, C.c_void_p(idx_out.data_ptr()),
                                                      C.c_void_p(dist_out.data_ptr()) if dist_out is not None else None, stream))

    def project_into(self, X, Y_out, which: int = _lib.PROJECT_LAMBDA):
        """Device path of project / project_omega: X and Y_out are torch tensors on the GPU (N x D -> N x D or N x M)."""
        import torch

        self._check_torch(X)
        width = self.M if which == _lib.PROJECT_OMEGA else self.D
        if tuple(Y_out.shape) != (X.shape[0], width) or Y_out.dtype != torch.float32 or not Y_out.is_contiguous():
            raise ValueError("Y_out must be a contiguous float32 N x %d tensor" % width)
        stream = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
        self._chk(self._L.vqhip_project_device(self._h, C.c_void_p(X.data_ptr()), X.shape[0], int(which),
                                               C.c_void_p(Y_out.data_ptr()), stream))

    def _assign_torch(self, X, y, want_dist):
        import torch

        self._check_torch(X)
        N = X.shape[0]
        idx = torch.empty(N, dtype=torch.int32, device=X.device)
        dist = torch.empty(N, dtype=torch.float32, device=X.device) if want_dist else None
        if y is None:
            self.assign_into(X, idx, dist)
            return idx, dist
        if not (y.is_cuda and y.dtype == torch.int32 and y.is_contiguous() and y.shape == (N,)):
            raise ValueError("y must be a contiguous CUDA int32 tensor with one class id per sample")
        jn = torch.empty(N, dtype=torch.int32, device=X.device)
        dn = torch.empty(N, dtype=torch.float32, device=X.device)
        self.assign_into(X, idx, dist, y, jn, dn)
        return idx, dist, jn, dn

    def _check_outputs(self, N, idx_out, dist_out=None, jneg_out=None, dneg_out=None):
        """Output tensors are written by the kernels through raw pointers: N elements, 4 bytes each, contiguous,
        on the context's GPU."""
        import torch

        for t, dt, name in ((idx_out, torch.int32, "idx_out"), (dist_out, torch.float32, "dist_out"),
                            (jneg_out, torch.int32, "jneg_out"), (dneg_out, torch.float32, "dneg_out")):
            if t is None:
                if name == "idx_out":
                    raise ValueError("idx_out is required")
                continue
            if not (t.is_cuda and t.device.index == self.device and t.dtype == dt and t.is_contiguous() and t.numel() >= N
                    and t.dim() == 1):
                raise ValueError(f"{name} must be a contiguous 1-D CUDA {dt} tensor on cuda:{self.device} with at least {N} elements")

    def _check_shape(self, shape):
        if len(shape) != 2 or shape[1] != self.D:
            raise ValueError(f"X must be N x {self.D}")

    def _check_torch(self, X):
        import torch

        if not (X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()):
            raise ValueError("X must be a contiguous CUDA float32 tensor")
        if X.device.index != self.device:
            raise ValueError(f"X lives on cuda:{X.device.index}, the context on cuda:{self.device}")
        self._check_shape(tuple(X.shape))


class MultiContext:
    """Several GPUs of one node behind one handle (vqhip_multi_*): the model replicated, the samples of a call cut into
    contiguous shards, one host thread per device, results in the caller's arrays.  numpy in, numpy out."""

    def __init__(self, devices):
        self._L = _lib.lib()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._L.vqhip_multi_create(devs, len(devices), C.byref(h))
        if rc != 0:
            raise VqhipError(rc, self._L.vqhip_multi_last_error(None).decode())
        self._h = h
        self.D = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.vqhip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise VqhipError(rc, self._L.vqhip_multi_last_error(self._h).decode())

    @property
    def device_count(self):
        return self._L.vqhip_multi_device_count(self._h)

    def set_prototypes(self, W):
        W = np.ascontiguousarray(W, dtype=np.float32)
        self._chk(self._L.vqhip_multi_set_prototypes(self._h, _fp(W), W.shape[0], W.shape[1]))
        self.K, self.D = W.shape

    def set_labels(self, cls):
        cls = np.ascontiguousarray(cls, dtype=np.uint32)
        self._chk(self._L.vqhip_multi_set_labels(self._h, _up(cls), cls.shape[0]))

    def set_omega(self, om):
        om = np.ascontiguousarray(om, dtype=np.float32)
        self._chk(self._L.vqhip_multi_set_omega(self._h, _fp(om), om.shape[0], om.shape[1]))

    def set_local_omegas(self, om):
        om = np.ascontiguousarray(om, dtype=np.float32)
        self._chk(self._L.vqhip_multi_set_local_omegas(self._h, _fp(om), om.shape[0], om.shape[1], om.shape[2]))

    def assign(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        if X.ndim != 2 or X.shape[1] != self.D:
            raise ValueError(f"X must be N x {self.D}")
        idx, dist = np.empty(X.shape[0], dtype=np.uint32), np.empty(X.shape[0], dtype=np.float32)
        self._chk(self._L.vqhip_multi_assign(self._h, _fp(X), X.shape[0], _up(idx), _fp(dist)))
        return idx, dist

    def assign_rows(self, rows):
        rows = [np.ascontiguousarray(r, dtype=np.float64) for r in rows]
        for r in rows:
            if r.ndim != 1 or r.shape[0] != self.D:
                raise ValueError(f"every row must hold {self.D} values")
        N = len(rows)
        ptrs = (C.POINTER(C.c_double) * max(N, 1))(*[r.ctypes.data_as(C.POINTER(C.c_double)) for r in rows])
        idx, dist = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.float32)
        self._chk(self._L.vqhip_multi_assign_rows_f64(self._h, ptrs, N, self.D, _up(idx), _fp(dist)))
        return idx, dist

    def glvq_batch_gradient(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.uint32)
        dW = np.empty((self.K, self.D), dtype=np.float32)
        cost, used = C.c_double(0.0), C.c_uint64(0)
        self._chk(self._L.vqhip_multi_glvq_batch_gradient(self._h, _fp(X), _up(y), X.shape[0], _fp(dW), C.byref(cost), C.byref(used)))
        return dW, cost.value, used.value

    def assign_device_shards(self, shards, idx_all, dist_all=None):
        """vqhip_multi_assign_device: shards[g] = CUDA float32 tensor (n_g x D) on device g of the handle; idx_all[g] /
        dist_all[g] = int32 / float32 tensors of sum(n_g) elements on device g.  Every device ends up with all winners
        (RCCL all-gather in place).  Returns after the exchange has completed."""
        import torch

        n = self.device_count
        if len(shards) != n or len(idx_all) != n or (dist_all is not None and len(dist_all) != n):
            raise ValueError("one shard and one result array per device of the handle")
        total = sum(int(x.shape[0]) for x in shards)
        for g in range(n):
            x = shards[g]
            if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2 and x.shape[1] == self.D):
                raise ValueError(f"shard {g} must be a contiguous CUDA float32 n x {self.D} tensor")
            outs = [(idx_all[g], torch.int32)] + ([(dist_all[g], torch.float32)] if dist_all is not None else [])
            for t, dt in outs:
                if not (t.is_cuda and t.device == x.device and t.dtype == dt and t.is_contiguous() and t.dim() == 1 and t.numel() >= total):
                    raise ValueError(f"result arrays of device {g} must be contiguous 1-D tensors of {total} elements on the shard's device")
        torch.cuda.synchronize()                     # the library works on streams of its own
        vp = C.c_void_p * n
        xs = vp(*[C.c_void_p(x.data_ptr()) for x in shards])
        rows = (C.c_uint64 * n)(*[int(x.shape[0]) for x in shards])
        ii = vp(*[C.c_void_p(t.data_ptr()) for t in idx_all])
        dd = vp(*[C.c_void_p(t.data_ptr()) for t in dist_all]) if dist_all is not None else None
        self._chk(self._L.vqhip_multi_assign_device(self._h, xs, rows, ii, dd))

    def assign_matched(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.uint32)
        N = X.shape[0]
        jp, jn = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint32)
        dp, dn = np.empty(N, dtype=np.float32), np.empty(N, dtype=np.float32)
        self._chk(self._L.vqhip_multi_assign_matched(self._h, _fp(X), _up(y), N, _up(jp), _fp(dp), _up(jn), _fp(dn)))
        return jp, dp, jn, dn
