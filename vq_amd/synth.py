"""Seeded synthetic models and samples for the BASELINE.json configs.

All values come from the counter-based generator in libvqhip (vqhip_synth_fill_host /
_device), so any index range of a device-resident sample matrix can be regenerated on the
CPU bit for bit.  Seed 42 is the seed every reference test uses (e.g. tests/vq.rs:20).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

SEED = 42
STREAM_W, STREAM_X, STREAM_OMEGA, STREAM_LOCAL0 = 1, 2, 3, 16
NOISE = 0.25


@dataclass(frozen=True)
class Config:
    name: str
    model: str          # which reference model's predict this is
    N: int
    D: int
    K: int
    M: int = 0          # rows of Omega (0: Euclidean)
    omega: str = "none"  # none | global | local
    classes: int = 8


CONFIGS = {
    "c1": Config("c1", "vq::VQ", 150, 4, 3, classes=3),
    "c2": Config("c2", "GLVQ", 10_000_000, 128, 256),
    "c3": Config("c3", "GMLVQ", 10_000_000, 128, 256, M=128, omega="global"),
    "c4": Config("c4", "LiRaMLVQ", 100_000_000, 256, 1024, M=16, omega="global"),
    "c5": Config("c5", "LGMLVQ", 50_000_000, 128, 64, M=128, omega="local"),
    # small-K Euclidean streaming case (HBM-rate demonstration)
    "c2k32": Config("c2k32", "GLVQ", 10_000_000, 128, 32),
    "c2k8": Config("c2k8", "GLVQ", 10_000_000, 128, 8),
    # many-prototype Euclidean case (per-batch costs amortised over 128 tiles): kernel tuning only
    "xk4096": Config("xk4096", "GLVQ", 1_000_000, 128, 4096),
}


def values(count: int, first: int = 0, seed: int = SEED, stream: int = STREAM_X, scale: float = 1.0,
           base: np.ndarray | None = None) -> np.ndarray:
    """Host fill: out[i] = base[(first+i) % len(base)] + scale * value(seed, stream, first+i)."""
    out = np.empty(count, dtype=np.float32)
    b = np.ascontiguousarray(base, dtype=np.float32).ravel() if base is not None else None
    rc = _lib.lib().vqhip_synth_fill_host(
        out.ctypes.data_as(C.POINTER(C.c_float)), count, first, seed, stream, scale,
        b.ctypes.data_as(C.POINTER(C.c_float)) if b is not None else None, b.size if b is not None else 0)
    if rc != 0:
        raise _lib.VqhipError(rc, "vqhip_synth_fill_host")
    return out


def _normalize(om: np.ndarray) -> np.ndarray:
    """Omega / sqrt(trace(Omega^T Omega)) = Omega / ||Omega||_F  (what src/gmlvq/gmlvq.rs:112-122 does)."""
    om = om.astype(np.float64)
    return om / np.sqrt((om * om).sum())


@dataclass
class Model:
    cfg: Config
    W: np.ndarray                  # K x D float32
    cls: np.ndarray                # K uint32
    omega: np.ndarray | None       # M x D or K x M x D float32


def make_model(cfg: Config, seed: int = SEED) -> Model:
    W = values(cfg.K * cfg.D, seed=seed, stream=STREAM_W).reshape(cfg.K, cfg.D)
    cls = (np.arange(cfg.K) % cfg.classes).astype(np.uint32)
    omega = None
    if cfg.omega == "global":
        u = values(cfg.M * cfg.D, seed=seed, stream=STREAM_OMEGA).reshape(cfg.M, cfg.D)
        if cfg.M == cfg.D:        # GMLVQ-like: near identity (the reference starts at I/sqrt(n), gmlvq.rs:160-161)
            omega = _normalize(np.eye(cfg.D) + 0.1 * u)
        else:                     # LiRaMLVQ: U(-1,1) M x D, normalised (liramlvq.rs:167-171)
            omega = _normalize(u)
        omega = omega.astype(np.float32)
    elif cfg.omega == "local":
        oms = []
        for j in range(cfg.K):
            u = values(cfg.M * cfg.D, seed=seed, stream=STREAM_LOCAL0 + j).reshape(cfg.M, cfg.D)
            oms.append(_normalize(np.eye(cfg.M, cfg.D) + 0.1 * u))
        omega = np.stack(oms).astype(np.float32)
    return Model(cfg, W, cls, omega)


def samples_host(model: Model, first: int, count: int, seed: int = SEED) -> np.ndarray:
    """Rows [first, first+count) of the sample matrix: X[i] = W[i mod K] + 0.25 * u(seed, 2, i*D + d)."""
    D = model.cfg.D
    return values(count * D, first=first * D, seed=seed, stream=STREAM_X, scale=NOISE, base=model.W).reshape(count, D)


def sample_labels(model: Model, first: int, count: int) -> np.ndarray:
    """y[i] = cls[i mod K]."""
    return model.cls[(np.arange(first, first + count) % model.cfg.K)].astype(np.uint32)


def samples_device(ctx, model: Model, first: int, count: int, seed: int = SEED):
    """Same rows generated on the GPU into a new torch tensor (count x D, float32)."""
    import torch

    D = model.cfg.D
    dev = torch.device("cuda", ctx.device)
    X = torch.empty((count, D), dtype=torch.float32, device=dev)
    base = torch.from_numpy(model.W).to(dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = ctx._L.vqhip_synth_fill_device(ctx._h, C.c_void_p(X.data_ptr()), count * D, first * D, seed, STREAM_X,
                                        NOISE, C.c_void_p(base.data_ptr()), base.numel(), stream)
    ctx._chk(rc)
    torch.cuda.current_stream(dev).synchronize()      # `base` may be freed after return
    return X


def load_model(ctx, model: Model, with_labels: bool = True):
    ctx.set_prototypes(model.W)
    if with_labels:
        ctx.set_labels(model.cls)
    if model.cfg.omega == "global":
        ctx.set_omega(model.omega)
    elif model.cfg.omega == "local":
        ctx.set_local_omegas(model.omega)
