"""Scratch timing of arbitrary shapes through the device path: python tools/shape_bench.py D K [M] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vq_amd import Context
D, K = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4_000_000
rng = np.random.default_rng(0)
ctx = Context(0)
if os.environ.get("SHAPE_BENCH_SCALAR"):
    from vq_amd import KERNEL_SCALAR
    ctx.set_kernel_policy(KERNEL_SCALAR)
ctx.set_prototypes(rng.uniform(-1, 1, size=(K, D)).astype(np.float32))
if M and os.environ.get("SHAPE_BENCH_LOCAL"):
    ctx.set_local_omegas((rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3)).astype(np.float32))
elif M:
    ctx.set_omega((rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3)).astype(np.float32))
X = torch.empty(N, D, device="cuda").uniform_(-1, 1)
idx = torch.empty(N, dtype=torch.int32, device="cuda"); dist = torch.empty(N, dtype=torch.float32, device="cuda")
for _ in range(2): ctx.assign_into(X, idx, dist)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): ctx.assign_into(X, idx, dist)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"D={D} K={K} M={M} kernel={ctx.last_kernel} N={N} ms={ms:.3f} samples/s={N/ms*1e3:.4g} GB/s={N*(D*4+8)/ms/1e6:.0f}")
if M and os.environ.get("SHAPE_BENCH_PROJECT"):
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    for which, width, nm in ((PROJECT_LAMBDA, D, "project"), (PROJECT_OMEGA, M, "project_omega")):
        Y = torch.empty((N, width), dtype=torch.float32, device="cuda")
        for _ in range(2): ctx.project_into(X, Y, which)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3): ctx.project_into(X, Y, which)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"D={D} M={M} {nm} kernel={ctx.last_kernel} N={N} ms={ms:.3f} samples/s={N/ms*1e3:.4g} GB/s={N*(D+width)*4/ms/1e6:.0f}")
