#!/usr/bin/env python3
"""bench.py -- samples/s of nearest-prototype assignment (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over this rank's resident synthetic shard (inputs already in
HBM): N=1 runs BASELINE.json configs[1], "GLVQ predict, synthetic 10M x 128 fp32, 256 prototypes".
With N ranks every rank holds its own 10M x 128 shard (weak scaling), prototypes replicated, and
the only exchange is the RCCL all-gather of winner indices + distances.  It runs on a side stream
out of one of two alternating result buffers, so the gather of pass k overlaps the kernel of pass
k+1 (the kernel claims its batches from a queue, so CUs the gather occupies at launch cost only
their share of the work).  All gathers have completed before the timed region ends.

One JSON line on rank 0: the contract fields plus
  roofline     dominant kernel (k_resident_pipe for c2, k_omega_pipe c3, k_lowrank_pipe c4, k_engine c5) against the f32 MFMA peak: algorithmic flops per launch
               (2*K*D per sample) / mean launch time from HIP events on the launch stream
  cpu_baseline the f64 oracle (a C port of the reference's loops; the Rust reference cannot be built
               here) timed on a bounded sample on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA, dense
PEAK_HBM_GBS = 8000.0            # spec; ~6300 achievable
# device function behind each kernel family name reported by vqhip_last_kernel
KERNEL_FN = {"mfma_euclid_lds_pipe": "k_resident_pipe", "mfma_euclid_lds_staged": "k_resident_pipe", "mfma_euclid_lds": "k_resident", "mfma_omega_lds": "k_omega_pipe",
             "mfma_omega_lowrank_lds": "k_lowrank_pipe", "scalar": "k_assign_scalar"}


def flops_per_sample(cfg):
    if cfg.omega == "none":
        return 2 * cfg.K * cfg.D                              # GEMM form of ||x - w||^2 screening
    if cfg.omega == "global":
        return 2 * cfg.M * cfg.D + 2 * cfg.K * cfg.M          # projection + search in the projected space
    return cfg.K * (2 * cfg.M * cfg.D + 3 * cfg.M)            # K local projections + squared norms


def bytes_per_sample(cfg):
    return cfg.D * 4 + 8                                      # sample row in, u32 index + f32 distance out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", help="config name in vq_amd.synth.CONFIGS")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the config's N, capped at 10M)")
    ap.add_argument("--gather-chunks", type=int, default=1, help="chunks per step when N > 1 (finer gather overlap)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the gather path (side stream, events, RCCL all_gather) even with one rank: rehearsal on a 1-GPU box")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that write banners to fd 1 (RCCL prints its version
    # there at communicator set-up) are pointed at stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from vq_amd import Context, synth
    from vq_amd.dist import gather_chunk, make_sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather = world > 1 or args.force_gather
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    cfg = synth.CONFIGS[args.workload]
    rows = args.rows or min(cfg.N, 10_000_000)
    sh = make_sharding(rows, world, rank, 1 if world == 1 else args.gather_chunks)
    chunks, m_rows, rows = sh.chunks, sh.rows_per_chunk, sh.rows_per_rank

    ctx = Context(local_rank)
    model = synth.make_model(cfg)
    synth.load_model(ctx, model)

    # Block-cyclic sharding: chunk c of rank r = global rows [(c*world + r)*m, +m), so the all-gather of
    # chunk c lands as the contiguous global range [c*world*m, (c+1)*world*m): natural order, no copies.
    X = [synth.samples_device(ctx, model, sh.first_row(c), m_rows) for c in range(chunks)]
    # two result buffers per chunk, alternating by pass: the gather of pass k reads one while pass k+1 writes the other
    idx_buf = [[torch.empty(m_rows, dtype=torch.int32, device=dev) for _ in range(chunks)] for _ in range(2)]
    dst_buf = [[torch.empty(m_rows, dtype=torch.float32, device=dev) for _ in range(chunks)] for _ in range(2)]
    idx_loc, dst_loc = idx_buf[0], dst_buf[0]
    if gather:
        idx_all = torch.empty(sh.total_rows, dtype=torch.int32, device=dev)
        dst_all = torch.empty(sh.total_rows, dtype=torch.float32, device=dev)
        comm = torch.cuda.Stream(device=dev)
        computed = [torch.cuda.Event() for _ in range(chunks)]
        gathered = [[torch.cuda.Event() for _ in range(chunks)] for _ in range(2)]
    compute = torch.cuda.current_stream(dev)
    kernel_ms = []                                   # (start, end) event pairs around every launch

    passes = [0]

    def step(timed):
        nonlocal idx_loc, dst_loc
        par = passes[0] & 1
        passes[0] += 1
        idx_loc, dst_loc = idx_buf[par], dst_buf[par]
        for c in range(chunks):
            if gather:
                compute.wait_event(gathered[par][c])     # the gather two passes ago has read this buffer
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(compute)
            ctx.assign_into(X[c], idx_loc[c], dst_loc[c])
            if timed:
                e1.record(compute)
                kernel_ms.append((e0, e1))
            if gather:
                computed[c].record(compute)
                with torch.cuda.stream(comm):
                    comm.wait_event(computed[c])
                    gather_chunk(sh, c, idx_all, dst_all, idx_loc[c], dst_loc[c])
                    gathered[par][c].record(comm)

    def fence():
        if gather:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if gather:
        for par in range(2):
            for e in gathered[par]:
                e.record(comm)
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: structured data => sample g belongs to prototype g mod K
    if cfg.omega == "none":
        for c in range(chunks):
            g0 = sh.first_row(c)
            want = (torch.arange(g0, g0 + m_rows, device=dev) % cfg.K).to(torch.int32)
            assert bool((idx_loc[c] == want).all()), "assignment of the timed pass is wrong"
        if gather:
            want = (torch.arange(sh.total_rows, device=dev) % cfg.K).to(torch.int32)
            assert bool((idx_all == want).all()), "gathered assignment is wrong"

    launch_ms = float(np.mean([a.elapsed_time(b) for a, b in kernel_ms]))
    if rank == 0:
        total_rows = rows * world
        fl = flops_per_sample(cfg)
        achieved = m_rows * fl / (launch_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "samples/sec nearest-prototype assignment",
            "value": total_rows * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg.model} predict, synthetic {rows}x{cfg.D} f32 per GPU, {cfg.K} prototypes"
                                   + (f", Omega {cfg.M}x{cfg.D} ({cfg.omega})" if cfg.omega != "none" else ""),
                       "name": cfg.name, "rows_per_gpu": rows, "dim": cfg.D, "prototypes": cfg.K,
                       "gather_chunks": chunks, "kernel": ctx.last_kernel,
                       "exchange": "rccl all_gather(idx u32, dist f32)" if gather else "none"},
            "roofline": {"bound": "mfma", "kernel": KERNEL_FN.get(ctx.last_kernel, "k_engine"), "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "launch_ms": launch_ms, "flops_per_sample": fl, "samples_per_launch": m_rows,
                         "hbm_GBps_algorithmic": m_rows * bytes_per_sample(cfg) / (launch_ms * 1e-3) / 1e9,
                         "hbm_frac_of_8TBps": m_rows * bytes_per_sample(cfg) / (launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        cpu_rows = args.cpu_rows
        if cpu_rows < 0:
            # ~15 s of single-core work: K*D multiply-adds per sample (Euclidean), K*(M*D*D + D*D) when
            # Lambda is rebuilt for every (sample, prototype) pair like src/helpers.rs:38
            per_sample = cfg.K * cfg.D if cfg.omega == "none" else cfg.K * (cfg.M * cfg.D * cfg.D + cfg.D * cfg.D)
            cpu_rows = 0 if world > 1 else max(16, int(4.5e10 / per_sample))
        if cpu_rows:
            import oracle                                   # the checker, used here only as the timed CPU baseline

            cpu_rows = min(cpu_rows, m_rows)
            Xs = synth.samples_host(model, sh.first_row(0), cpu_rows)
            t1 = time.perf_counter()
            o_idx, _, _ = oracle.assign(Xs, model.W, omega=model.omega, hoist=False, threads=1)
            cpu_s = time.perf_counter() - t1
            same = bool((o_idx.view(np.int32) == idx_loc[0][:cpu_rows].cpu().numpy()).all())
            line["cpu_baseline"] = {"value": cpu_rows / cpu_s, "unit": "samples/s", "cores": 1, "kind": "port",
                                    "sample": f"first {cpu_rows} rows of the same shard, f64 oracle, faithful mode "
                                              f"(Lambda rebuilt per pair where Omega is used), {cpu_s:.1f} s",
                                    "indices_equal_gpu": same}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if gather:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
