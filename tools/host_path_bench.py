"""PCIe-inclusive rate of the host-pointer entry points (never bench.py's `value`):
python tools/host_path_bench.py [rows]   -- c2 shape, host arrays in, host arrays out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vq_amd import Context, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cfg = synth.CONFIGS["c2"]
ctx = Context(0)
m = synth.make_model(cfg)
synth.load_model(ctx, m)
X = synth.samples_host(m, 0, N)
ctx.assign(X[:100000])                                    # staging allocation + warm-up
t = time.perf_counter(); idx, dist = ctx.assign(X); dt = time.perf_counter() - t
print(f"vqhip_assign           (contiguous f32 host rows): {N/dt:.4g} samples/s, {N*(cfg.D*4+8)/dt/1e9:.1f} GB/s over the host link")
assert (idx == np.arange(N) % cfg.K).all()
n2 = min(N, 1_000_000)
rows = [np.array(r, dtype=np.float64) for r in X[:n2]]    # the reference's container: one allocation per row
t = time.perf_counter(); i2, d2 = ctx.assign_rows(rows); dt = time.perf_counter() - t
print(f"vqhip_assign_rows_f64  ({n2} separate f64 rows, incl. ctypes pointer table): {n2/dt:.4g} samples/s")
assert (i2 == idx[:n2]).all()
ctx.close()
