"""Latency of small device-resident assign calls (serving-sized batches), config 2's model: us per call, back to back on one stream.
    python tools/small_batch_bench.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vq_amd import Context, synth

rows = [int(a) for a in sys.argv[1:]] or [256, 2048, 8192, 32768, 131072, 524288]
ctx = Context(0)
m = synth.make_model(synth.CONFIGS["c2"])
synth.load_model(ctx, m)
for n in rows:
    X = synth.samples_device(ctx, m, 0, n)
    idx = torch.empty(n, dtype=torch.int32, device="cuda"); dist = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(5): ctx.assign_into(X, idx, dist)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 200
    ev[0].record()
    for _ in range(reps): ctx.assign_into(X, idx, dist)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / reps * 1e3
    ctx.prepare()
    for _ in range(5): ctx.assign_into_noalloc(X, idx, dist)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps): ctx.assign_into_noalloc(X, idx, dist)
    ev[1].record(); torch.cuda.synchronize()
    us1 = ev[0].elapsed_time(ev[1]) / reps * 1e3
    print(f"rows={n:8d}  assign_device {us:8.1f} us/call ({n / us:8.1f} M samples/s)   noalloc (one launch) {us1:8.1f} us/call", flush=True)
ctx.close()
