"""Scratch timing script (not part of the contract): python tools_quick_bench.py c2 [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vq_amd import Context, synth
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = synth.CONFIGS[name]
N = int(sys.argv[2]) if len(sys.argv) > 2 else min(cfg.N, 10_000_000)
ctx = Context(0)
m = synth.make_model(cfg)
synth.load_model(ctx, m)
X = synth.samples_device(ctx, m, 0, N)
idx = torch.empty(N, dtype=torch.int32, device="cuda"); dist = torch.empty(N, dtype=torch.float32, device="cuda")
for _ in range(2): ctx.assign_into(X, idx, dist)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
K = 5
ev[0].record()
for _ in range(K): ctx.assign_into(X, idx, dist)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / K
flops = {"none": 2*cfg.K*cfg.D, "global": 2*cfg.M*cfg.D + 2*cfg.K*cfg.M, "local": cfg.K*(2*cfg.M*cfg.D + 3*cfg.M)}[cfg.omega]
print(f"{name} kernel={ctx.last_kernel} N={N} ms={ms:.3f} samples/s={N/ms*1e3:.4g} TFLOP/s={N*flops/ms/1e9:.1f} GB/s={N*(cfg.D*4+8)/ms/1e6:.0f}")
ok = (idx.cpu().numpy() == np.arange(N) % cfg.K).mean()
print("frac idx == i mod K:", ok)
if len(sys.argv) > 3 and sys.argv[3] == "matched":
    y = torch.from_numpy(synth.sample_labels(m, 0, N).view(np.int32)).cuda()
    jn = torch.empty(N, dtype=torch.int32, device="cuda"); dn = torch.empty(N, dtype=torch.float32, device="cuda")
    for _ in range(2): ctx.assign_into(X, idx, dist, y=y, jneg_out=jn, dneg_out=dn)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(K): ctx.assign_into(X, idx, dist, y=y, jneg_out=jn, dneg_out=dn)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / K
    print(f"{name} matched (both searches) kernel={ctx.last_kernel} N={N} ms={ms:.3f} samples/s={N/ms*1e3:.4g} TFLOP/s={N*flops/ms/1e9:.1f}")
if len(sys.argv) > 3 and sys.argv[3] == "project" and cfg.omega == "global":
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    for which, width, fl, nm in ((PROJECT_LAMBDA, cfg.D, 2 * cfg.D * cfg.D, "project(Lambda)"), (PROJECT_OMEGA, cfg.M, 2 * cfg.M * cfg.D, "project_omega")):
        Y = torch.empty((N, width), dtype=torch.float32, device="cuda")
        for _ in range(2): ctx.project_into(X, Y, which)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(K): ctx.project_into(X, Y, which)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / K
        print(f"{name} {nm} kernel={ctx.last_kernel} N={N} ms={ms:.3f} samples/s={N/ms*1e3:.4g} TFLOP/s={N*fl/ms/1e9:.1f} GB/s={N*(cfg.D+width)*4/ms/1e6:.0f}")
if len(sys.argv) > 3 and sys.argv[3] == "project" and cfg.omega == "local":
    from vq_amd import PROJECT_LAMBDA
    Y = torch.empty((N, cfg.D), dtype=torch.float32, device="cuda")
    for _ in range(2): ctx.project_into(X, Y, PROJECT_LAMBDA)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(K): ctx.project_into(X, Y, PROJECT_LAMBDA)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / K
    print(f"{name} project(Lambda_j of the winner) kernel={ctx.last_kernel} N={N} ms={ms:.3f} samples/s={N/ms*1e3:.4g}")
