"""Scratch check: the labelled search under VQHIP_KERNEL_AUTO against the scalar kernels on the device, over a grid of
shapes of all model kinds (no oracle involved).  python tools/matched_grid_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import KERNEL_AUTO, KERNEL_SCALAR, Context

rng = np.random.default_rng(11)
ctx = Context(0)
N, bad, kernels = 400, 0, {}
cases = [(D, K, 0, "none") for D in (3, 16, 64, 100, 128, 200, 256, 300, 784) for K in (1, 2, 5, 33, 257, 1000)]
cases += [(D, K, M, "global") for D in (32, 100, 128, 200, 256, 300) for M in (2, 16, 40, 128, 200) for K in (3, 40, 300)]
cases += [(D, K, M, "local") for D in (16, 128, 200, 300) for M in (1, 33) for K in (2, 20)]
for D, K, M, kind in cases:
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    out = {}
    Om = None
    if kind == "global":
        Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    if kind == "local":
        Om = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    for policy in (KERNEL_SCALAR, KERNEL_AUTO):
        ctx.set_kernel_policy(policy)
        ctx.set_prototypes(W)
        if kind == "global":
            ctx.set_omega(Om)
        if kind == "local":
            ctx.set_local_omegas(Om)
        ctx.set_labels(cls)
        out[policy] = ctx.assign_matched(X, y) + (ctx.last_kernel,)
    s, a = out[KERNEL_SCALAR], out[KERNEL_AUTO]
    kernels[a[4]] = kernels.get(a[4], 0) + 1
    for name, si, sd, ai, ad in (("pos", s[0], s[1], a[0], a[1]), ("neg", s[2], s[3], a[2], a[3])):
        sd64, ad64 = sd.astype(np.float64), ad.astype(np.float64)
        fin = np.isfinite(sd64) & np.isfinite(ad64)
        inf_ok = (np.isfinite(sd64) == np.isfinite(ad64)).all()
        close = np.all(np.abs(ad64[fin] - sd64[fin]) <= 1e-4 * np.abs(sd64[fin]) + 2e-6)
        same = (si == ai).mean()
        if not (inf_ok and close and same > 0.97):
            bad += 1
            print("MISMATCH", (D, K, M, kind), name, a[4], "same", same, "inf_ok", inf_ok, "close", close)
print("cases", len(cases), "bad", bad, "kernels", kernels)
