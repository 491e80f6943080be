#!/usr/bin/env python3
"""bench.py -- samples/s of nearest-prototype assignment (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W            # starts its own N rank processes when N > 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     # or under a launcher (RANK / WORLD_SIZE set)

A step = one pass of the hot path over this rank's resident synthetic shard (inputs already in HBM).  N=1 runs
BASELINE.json configs[1], "GLVQ predict, synthetic 10M x 128 fp32, 256 prototypes".  One process per GPU; every sample
is independent (src/glvq/glvq.rs:224-232), so samples are sharded, the model is replicated, and the only exchange is
the RCCL all-gather of winner indices + distances.  It runs on a side stream out of one of two alternating result
buffers, so the gather of pass k overlaps the kernel of pass k+1.  All gathers have completed before the timed region
ends.
  --scaling weak    (default) every rank holds min(config N, 10M) rows: per-GPU work fixed
  --scaling strong  the config's own N split over the ranks, ceil(N / world) rows each: configs[3] (c4, 100M x 256) and
                    configs[4] (c5, 50M x 128) are fixed-size jobs

One JSON line on rank 0: the contract fields plus
  roofline     the dominant kernel against the roofline that bounds it: algorithmic flops (or bytes) per launch
                / mean launch time from HIP events on the launch stream; `traffic` = HBM bytes per launch
               from the PMC counters, collected live by two child runs under rocprofv3 (N=1 only; null when that fails)
  cpu_baseline the f64 oracle (a C port of the reference's loops; the Rust reference cannot be built here) timed on
               bounded samples on this box's host cores, rank 0, N=1 only: faithful 1 core (Lambda rebuilt per pair like
               src/helpers.rs:38), hoisted-Lambda 1 core, hoisted all cores
  extra        N=1 only: the other BASELINE configs (c3, a c4 shard, a c5 shard) and the HBM-leaning Euclidean cases
               (c2k32: 32 prototypes on the matrix cores; c2k8: 8 prototypes, the direct-form kernel), each timed the same way
               with its own roofline block
"""
import argparse
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6   # MI355X_MICROARCH.md: bf16-input MFMA, dense (32 cycles per 32x32x16 on 1 024 SIMDs at 2.4 GHz)
PEAK_HBM_GBS = 8000.0            # spec; ~6300 achievable
# device function behind each kernel family name reported by vqhip_last_kernel
KERNEL_FN = {"mfma_euclid_lds_pipe": "k_resident_pipe", "mfma_euclid_lds_staged": "k_resident_pipe",
             "mfma_euclid_lds_pipe_cut": "k_resident_pipe", "mfma_euclid_lds": "k_resident", "mfma_omega_lds": "k_omega_pipe",
             "mfma_omega_lowrank_lds": "k_lowrank_pipe", "mfma_euclid_long": "k_long", "mfma_omega_long": "k_omega_long",
             "mfma_local_long": "k_local_long", "mfma_local_bf16": "k_local_pj", "valu_euclid_direct": "k_direct", "scalar": "k_assign_scalar"}
HBM_BOUND = {"c2k32", "c2k8"}    # few prototypes: the memory pipe binds, not the matrix pipe


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", help="config name in vq_amd.synth.CONFIGS")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (overrides --scaling's choice)")
    ap.add_argument("--gather-chunks", type=int, default=1, help="chunks per step when N > 1 (finer gather overlap)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the gather path (side stream, events, RCCL all_gather) even with one rank: rehearsal on a 1-GPU box")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the faithful CPU baseline sample (0 = no CPU baseline)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs (N=1 runs them by default)")
    ap.add_argument("--traffic", choices=("live", "off"), default="live",
                    help="live: two child runs under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE) for roofline.traffic, N=1 only")
    ap.add_argument("--standin", action="store_true",
                    help="no GPU: ranks on the CPU over gloo, the assignment replaced by the known answer of the structured "
                         "data (row g -> prototype g mod K).  Exercises launcher, sharding, gather and the JSON line only")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ the launcher (no GPU call in this process)

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes (this process has made no
    GPU call and makes none), wait for them, pass rank 0's JSON line through on the shared stdout."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:                       # one rank failed: the others would wait in a collective for ever
                        try:
                            os.killpg(q.pid, signal.SIGTERM)
                        except ProcessLookupError:
                            pass
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    return rc


# ------------------------------------------------------------------ per-sample work

def flops_per_sample(cfg):
    if cfg.omega == "none":
        return 2 * cfg.K * cfg.D                              # GEMM form of ||x - w||^2 screening
    if cfg.omega == "global":
        return 2 * cfg.M * cfg.D + 2 * cfg.K * cfg.M          # projection + search in the projected space
    return cfg.K * (2 * cfg.M * cfg.D + 3 * cfg.M)            # K local projections + squared norms


def bytes_per_sample(cfg):
    return cfg.D * 4 + 8                                      # sample row in, u32 index + f32 distance out


def matrix_work(cfg, ranking, projection="f32"):
    """Per sample: (flops the f32 matrix cores execute, flops the bf16 matrix cores execute).  With ranking "bf16x3" the
    search products (2 K L of the algorithmic flops) run as three bf16 piece products each; with projection "bf16x6"
    the projection's products (the rest of an Omega model's flops) run as six."""
    fl = flops_per_sample(cfg)
    search = 2 * cfg.K * (cfg.D if cfg.omega == "none" else cfg.M) if cfg.omega != "local" else 0
    proj = fl - search
    f32_fl, bf_fl = fl, 0
    if ranking == "bf16x3":
        f32_fl -= search
        bf_fl += 3 * search
    if projection == "bf16x6":
        f32_fl -= proj
        bf_fl += 6 * proj
    return f32_fl, bf_fl


# What the SQ counters of profiles/ say binds a launch when it is not the roofline it is priced against:
# the bf16-ranked kernels keep the matrix pipe about half busy -- vector issue (screening, epilogue) and the L2 / HBM
# round trips of the batch epilogue fill the rest; the f32 kernels sit on the matrix pipe; few prototypes stream.
def limiter_of(cfg, ranking, projection):
    if cfg.name in HBM_BOUND:
        return "hbm stream"
    if ranking == "bf16x3" or projection == "bf16x6":
        return "valu-issue + epilogue latency"
    return "mfma"


def roofline_block(cfg, kernel, rows, launch_ms, traffic=None, traffic_source=None, ranking="f32", projection="f32"):
    """achieved = ALGORITHMIC flops (or bytes) per launch / launch time.  MFMA peak = the same algorithmic flops over the
    least time the matrix cores need for the instructions this kernel issues for them: f32 products at 157.3 TFLOP/s and,
    where the candidates are ranked on the bf16 cores, three bf16 piece products per algorithmic product at 2 516.6
    (so 838.9 algorithmic TFLOP/s for a pure search), six per projection product.  `bound` names the roofline the
    fraction is priced against (hbm | mfma), `limiter` what the counters say binds the launch."""
    fl, by = flops_per_sample(cfg), bytes_per_sample(cfg)
    secs = launch_ms * 1e-3
    tflops = rows * fl / secs / 1e12
    gbs = rows * by / secs / 1e9
    f32_fl, bf_fl = matrix_work(cfg, ranking, projection)
    t_min = f32_fl / (PEAK_F32_MFMA_TFLOPS * 1e12) + bf_fl / (PEAK_BF16_MFMA_TFLOPS * 1e12)       # per sample
    mfma_peak = fl / t_min / 1e12
    blk = {}
    if cfg.name in HBM_BOUND:
        blk.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
    else:
        blk.update({"bound": "mfma", "achieved": tflops, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tflops / mfma_peak})
    blk.update({"traffic": traffic, "traffic_over_algorithmic": traffic / (rows * by) if traffic else None,
                "traffic_source": traffic_source, "limiter": limiter_of(cfg, ranking, projection),
                "kernel": KERNEL_FN.get(kernel, "k_engine"), "ranking": ranking, "projection": projection,
                "launch_ms": launch_ms, "samples_per_launch": rows, "flops_per_sample": fl, "bytes_per_sample": by,
                "tflops": tflops, "frac_of_f32_mfma_157.3": tflops / PEAK_F32_MFMA_TFLOPS,
                "hbm_GBps": gbs, "hbm_frac_of_8TBps": gbs / PEAK_HBM_GBS})
    return blk


def sig4(v):
    """Floats to 4 significant digits, recursively: the JSON line has to survive the driver's tail."""
    if isinstance(v, float):
        return float(f"{v:.4g}")
    if isinstance(v, dict):
        return {k: sig4(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [sig4(x) for x in v]
    return v


# ------------------------------------------------------------------ HBM traffic from the PMC counters, live

def measure_traffic(workload, rows, kernel_fn):
    """HBM bytes per launch of the dominant kernel: FETCH_SIZE and WRITE_SIZE in one rocprofv3 --pmc pass each (they do
    not fit one pass), each pass a fresh child run of this script; MI355X_MICROARCH.md (HBM): the counters are in KiB,
    FETCH_SIZE reports half of a wide coalesced read stream on gfx950 and is doubled, WRITE_SIZE is exact."""
    import csv
    import glob

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="vqhip_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload, "--rows", str(rows), "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--no-extras",
               "--traffic", "off"]
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                return None, f"rocprofv3 --pmc {counter} timed out"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {p.returncode})"
            by_name = {}                       # per instantiation: the finish pass is another one of the same template
            for f in files:
                for r in csv.DictReader(open(f)):
                    if ("vq::" + kernel_fn + "<") in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        by_name.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not by_name:
                return None, f"no {counter} rows for {kernel_fn}"
            # per step: every instantiation's launches count (first pass + finish pass), over the first pass's launches
            first = max(by_name.values(), key=sum)
            vals[counter] = sum(sum(v) for v in by_name.values()) / len(first)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "live pmc: 2*FETCH_SIZE+WRITE_SIZE KiB"


def stored_traffic(workload):
    """A committed PMC profile of this workload (profiles/pmc_latest.json), named as the source."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get(workload)
        if e:
            return e["hbm_bytes_per_launch"], "profiles/" + e.get("from", "pmc_latest.json")
    except Exception:
        pass
    return None, None


# ------------------------------------------------------------------ CPU baselines (the oracle as the timed thing)

def cpu_baselines(cfg, model, first_row, faithful_rows, gpu_idx, budget_s=4.0):
    """(i) faithful single thread (Lambda rebuilt per (sample, prototype) pair, helpers.rs:38 -- the
    reference's real cost), (ii) hoisted Lambda single thread, (iii) hoisted over all host cores.  Each on a sample sized
    for about budget_s seconds; rates are per second of wall clock on that sample."""
    import numpy as np

    import oracle                                   # the checker, used here only as the timed CPU baseline
    from vq_amd import synth

    cores = min(oracle.max_threads(), os.cpu_count() or 1, 16)     # the GPU box gives one GPU 16 host cores

    def macs(hoist):                                 # multiply-adds per sample
        if cfg.omega == "none":
            return cfg.K * cfg.D
        lam = cfg.M * cfg.D * cfg.D                  # building Lambda = Omega^T Omega once
        return cfg.K * ((0 if hoist else lam) + cfg.D * cfg.D + cfg.D)

    def run(rows, hoist, threads):
        rows = max(4, int(rows))
        Xs = synth.samples_host(model, first_row, rows)
        t = time.perf_counter()
        idx, _, _ = oracle.assign(Xs, model.W, omega=model.omega, hoist=hoist, threads=threads)
        dt = time.perf_counter() - t
        return rows, dt, idx

    rate1 = 3e9 if cfg.omega == "none" else 6e8      # multiply-adds per second and core, -O2 scalar f64 (measured order of magnitude)
    out = {}
    rows, dt, idx = run(faithful_rows if faithful_rows > 0 else budget_s * rate1 / macs(False), False, 1)
    same = bool((idx.view(np.int32) == gpu_idx[:rows]).all()) if gpu_idx is not None else None
    out.update({"value": rows / dt, "unit": "samples/s", "cores": 1, "kind": "port",
                "sample": f"first {rows} rows of the shard, f64 C oracle as the reference computes (Lambda per pair), {dt:.1f} s",
                "indices_equal_gpu": same})
    if cfg.omega != "none":
        rows, dt, _ = run(budget_s * rate1 / macs(True), True, 1)
        out["hoisted_1core"] = rows / dt                 # Lambda built once per model, 1 core
    else:
        out["hoisted_1core"] = out["value"]              # no Omega: the faithful leg
    rows, dt, _ = run(budget_s * rate1 * cores * 0.7 / macs(True), True, cores)
    out["all_cores"] = {"value": rows / dt, "cores": cores}     # hoisted Lambda, threads over samples
    return out


# ------------------------------------------------------------------ one rank

def run_rank(args):
    # stdout carries exactly one JSON line: libraries that write banners to fd 1 (RCCL prints its version
    # there at communicator set-up) are pointed at stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from vq_amd import OPT_PROJECTION, OPT_RANKING, VALUE_DEFAULT, VALUE_F32, synth
    from vq_amd.dist import gather_chunk, make_sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    standin = args.standin
    if standin:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    gather = world > 1 or args.force_gather
    exchange = "none"
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if standin:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            exchange = f"gloo all_gather(idx u32, dist f32), {dist.get_world_size()} ranks (CPU stand-in)"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            exchange = f"rccl {ver} all_gather(idx u32, dist f32), {dist.get_world_size()} ranks over {dist.get_backend()}"

    cfg = synth.CONFIGS[args.workload]
    if args.rows:
        rows = args.rows
    elif args.scaling == "strong":
        rows = (cfg.N + world - 1) // world
    else:
        rows = min(cfg.N, 10_000_000)
    if rows * cfg.D * 4 > 200e9:
        raise SystemExit(f"{rows} rows x {cfg.D} f32 do not fit one GPU next to the results: use more GPUs or --rows")
    sh = make_sharding(rows, world, rank, 1 if world == 1 else args.gather_chunks)
    chunks, m_rows, rows = sh.chunks, sh.rows_per_chunk, sh.rows_per_rank

    model = synth.make_model(cfg)
    ctx = None
    if standin:
        X = [None] * chunks
    else:
        from vq_amd import Context
        ctx = Context(local_rank)
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
    if gather and not standin:
        comm = torch.cuda.Stream(device=dev)
        computed = [torch.cuda.Event() for _ in range(chunks)]
        gathered = [[torch.cuda.Event() for _ in range(chunks)] for _ in range(2)]
    compute = None if standin else torch.cuda.current_stream(dev)
    kernel_ms = []                                   # (start, end) event pairs around every launch

    passes = [0]

    def assign(c, idx_out, dst_out):
        if standin:                                  # the known answer of the structured data, no kernel
            g0 = sh.first_row(c)
            idx_out.copy_((torch.arange(g0, g0 + m_rows) % cfg.K).to(torch.int32))
            dst_out.fill_(0.0)
        else:
            ctx.assign_into(X[c], idx_out, dst_out)

    def step(timed):
        nonlocal idx_loc, dst_loc
        par = passes[0] & 1
        passes[0] += 1
        idx_loc, dst_loc = idx_buf[par], dst_buf[par]
        for c in range(chunks):
            if standin:
                assign(c, idx_loc[c], dst_loc[c])
                if gather:
                    gather_chunk(sh, c, idx_all, dst_all, idx_loc[c], dst_loc[c])
                continue
            if gather:
                compute.wait_event(gathered[par][c])     # the gather two passes ago has read this buffer
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(compute)
            assign(c, idx_loc[c], dst_loc[c])
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
        if not standin:
            torch.cuda.synchronize(dev)

    if gather and not standin:
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

    if rank == 0:
        kernel = "stand-in (no kernel)" if standin else ctx.last_kernel
        line = {
            "metric": "samples/sec nearest-prototype assignment",
            "value": rows * world * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if not args.rows else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not standin else "synthetic, CPU stand-in for the kernel (launcher / gather rehearsal, not a measurement)",
            "config": {"workload": f"{cfg.model} predict, synthetic {rows}x{cfg.D} f32 per GPU, {cfg.K} prototypes"
                                   + (f", Omega {cfg.M}x{cfg.D} ({cfg.omega})" if cfg.omega != "none" else ""),
                       "name": cfg.name, "rows_per_gpu": rows, "total_rows": rows * world, "dim": cfg.D, "prototypes": cfg.K,
                       "gather_chunks": chunks, "kernel": kernel, "exchange": exchange},
        }
        if not standin:
            launch_ms = float(np.mean([a.elapsed_time(b) for a, b in kernel_ms]))
            traffic, source = None, None
            if world == 1 and args.traffic == "live":
                traffic, source = measure_traffic(args.workload, rows, KERNEL_FN.get(kernel, "k_engine"))
            if traffic is None and m_rows == min(cfg.N, 10_000_000):
                why = source
                traffic, source = stored_traffic(args.workload)
                if traffic is not None and why:
                    source += f" (live collection: {why})"
            line["roofline"] = roofline_block(cfg, kernel, m_rows, launch_ms, traffic, source, ctx.last_ranking, ctx.last_projection)
            line["config"]["ranking"] = ctx.last_ranking
            line["config"]["projection"] = ctx.last_projection
            if world == 1 and args.cpu_rows != 0:
                line["cpu_baseline"] = cpu_baselines(cfg, model, sh.first_row(0), args.cpu_rows if args.cpu_rows > 0 else 0,
                                                     idx_loc[0].cpu().numpy(), budget_s=6.0)
    # the other configs, one GPU: same timing, own roofline blocks
    if rank == 0 and world == 1 and not standin and not args.no_extras and args.workload == "c2":
        del X, idx_buf, dst_buf
        torch.cuda.empty_cache()
        line["extra"] = {}
        # c2_f32 / c3_f32 / c5_f32: the same workloads with everything on the f32 matrix cores (vqhip_set_option): the
        # kernels the f32-MFMA fractions of DESIGN.md section 5 are quoted on
        for name, xrows in (("c3", 10_000_000), ("c4", 10_000_000), ("c5", 4_000_000), ("c2k32", 10_000_000), ("c2k8", 10_000_000),
                            ("c2_f32", 10_000_000), ("c3_f32", 10_000_000), ("c5_f32", 4_000_000)):
            f32_only = name.endswith("_f32")
            xc = synth.CONFIGS[name[:-4] if f32_only else name]
            xm = synth.make_model(xc)
            for opt in (OPT_RANKING, OPT_PROJECTION):
                ctx.set_option(opt, VALUE_F32 if f32_only else VALUE_DEFAULT)
            synth.load_model(ctx, xm)
            ctx.prepare()
            Xx = synth.samples_device(ctx, xm, 0, xrows)
            xi = torch.empty(xrows, dtype=torch.int32, device=dev)
            xd = torch.empty(xrows, dtype=torch.float32, device=dev)
            for _ in range(3):
                ctx.assign_into(Xx, xi, xd)
            torch.cuda.synchronize(dev)
            evs = []
            n_launch = 20 if not name.startswith("c5") else 10
            tw = time.perf_counter()
            for _ in range(n_launch):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(compute)
                ctx.assign_into(Xx, xi, xd)
                e1.record(compute)
                evs.append((e0, e1))
            torch.cuda.synchronize(dev)
            wall = time.perf_counter() - tw
            ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
            traffic, source = stored_traffic(name) if not f32_only else (None, None)
            if traffic is not None and xrows != 10_000_000:     # every byte of these kernels' traffic is per batch of rows
                traffic, source = traffic * xrows / 10_000_000, source + f" (10M-row profile scaled to {xrows})"
            rb = roofline_block(xc, ctx.last_kernel, xrows, ms, traffic, source, ctx.last_ranking, ctx.last_projection)
            entry = {"shape": f"{xc.model} {xrows}x{xc.D} K={xc.K}" + (f" Om={xc.M}x{xc.D} {xc.omega}" if xc.omega != "none" else ""),
                     "value": xrows * n_launch / wall, "ms": wall / n_launch * 1e3, "launch_ms": ms, "kernel": ctx.last_kernel,
                     "rank": ctx.last_ranking, "proj": ctx.last_projection}
            entry.update({k: rb[k] for k in ("bound", "achieved", "peak", "unit", "frac", "limiter", "traffic_over_algorithmic",
                                             "frac_of_f32_mfma_157.3", "hbm_frac_of_8TBps")})
            if args.cpu_rows != 0 and not f32_only:
                cb = cpu_baselines(xc, xm, 0, 0, xi.cpu().numpy(), budget_s=2.5)
                entry["cpu_1core"] = cb["value"]
                entry["cpu_all_cores"] = cb["all_cores"]["value"]
                entry["cpu_idx_equal"] = cb["indices_equal_gpu"]
            line["extra"][name] = entry
            del Xx, xi, xd
            torch.cuda.empty_cache()
    if rank == 0:
        if "extra" in line:
            line["notes"] = ("extra: same timing, 20 launches (c5: 10, 4M rows); value = samples/s; peak of a bf16-ranked workload = algorithmic "
                             "flops over the matrix cores' least time for its instruction mix (3 bf16 products per search product, 6 per "
                             "projection product, f32 at 157.3); cpu_* = f64 C oracle on bounded samples (DESIGN.md 5)")
            # last key: the driver keeps the tail of the line
            # [samples/s, fraction of the roofline named next, that roofline (hbm | mfma), what the counters say binds]
            short = {"valu-issue + epilogue latency": "valu-issue", "hbm stream": "hbm", "mfma": "mfma"}
            line["extra_summary"] = {k: [e["value"], e["frac"], e["bound"], short.get(e.get("limiter"), e.get("limiter"))]
                                     for k, e in line["extra"].items()}
            line["extra_summary"][line["config"]["name"]] = [line["value"], line["roofline"]["frac"], line["roofline"]["bound"],
                                                             short.get(line["roofline"]["limiter"], line["roofline"]["limiter"])]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(sig4(line)) + "\n").encode())
    if gather:
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    have_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if have_launcher and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
    if args.gpus > 1 and not have_launcher:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
