"""CPU, world_size 2, gloo: the sharding + gather bookkeeping of the multi-GPU path
(vq_amd/dist.py).  The assignment itself is replaced by the known answer of the structured
synthetic data (row g -> prototype g mod K); the GPU kernels are covered by the -m gpu tests."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vq_amd.dist import gather_chunk, make_sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows_per_rank, chunks, K, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = make_sharding(rows_per_rank, world, rank, chunks)
        idx_all = torch.full((sh.total_rows,), -1, dtype=torch.int32)
        dst_all = torch.full((sh.total_rows,), -1.0, dtype=torch.float32)
        for c in range(sh.chunks):
            g = torch.arange(sh.first_row(c), sh.first_row(c) + sh.rows_per_chunk)
            gather_chunk(sh, c, idx_all, dst_all, (g % K).to(torch.int32), g.to(torch.float32) * 0.5)
        np.save(os.path.join(out_dir, f"idx{rank}.npy"), idx_all.numpy())
        np.save(os.path.join(out_dir, f"dst{rank}.npy"), dst_all.numpy())
    finally:
        dist.destroy_process_group()


def test_sharding_covers_every_row_once():
    for world, rows, chunks in ((1, 10, 1), (2, 10, 4), (8, 1000, 4), (4, 7, 3)):
        seen = []
        for r in range(world):
            sh = make_sharding(rows, world, r, chunks)
            assert sh.rows_per_rank >= rows
            for c in range(sh.chunks):
                seen.extend(range(sh.first_row(c), sh.first_row(c) + sh.rows_per_chunk))
        assert sorted(seen) == list(range(sh.total_rows))
        lo, hi = sh.gathered_range(sh.chunks - 1)
        assert hi == sh.total_rows and hi - lo == world * sh.rows_per_chunk


def test_two_rank_gather_is_in_natural_order(tmp_path):
    world, rows, chunks, K = 2, 1003, 4, 7
    mp.spawn(_worker, args=(world, _free_port(), rows, chunks, K, str(tmp_path)), nprocs=world, join=True)
    total = make_sharding(rows, world, 0, chunks).total_rows
    want_idx = (np.arange(total) % K).astype(np.int32)
    want_dst = np.arange(total, dtype=np.float32) * 0.5
    for r in range(world):
        assert (np.load(tmp_path / f"idx{r}.npy") == want_idx).all()
        assert (np.load(tmp_path / f"dst{r}.npy") == want_dst).all()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two rank processes itself (the parent makes no GPU
    call), they rendezvous on 127.0.0.1, shard, gather, and rank 0 prints exactly one JSON line.  Here on the CPU: gloo,
    and the kernel replaced by the structured data's known answer (--standin), which the run itself checks after the
    gather (row g -> prototype g mod K over the whole gathered array)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for extra in ([], ["--scaling", "strong", "--workload", "c1", "--gather-chunks", "3"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--standin", "--steps", "2", "--warmup", "1",
                            "--rows", "4001"] + extra, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, p.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 2 and out["unit"] == "samples/s" and out["value"] > 0
        assert "2 ranks" in out["config"]["exchange"] and "all_gather" in out["config"]["exchange"]
        assert out["config"]["total_rows"] == 2 * out["config"]["rows_per_gpu"]


def test_bench_strong_scaling_splits_the_configs_own_rows():
    """--scaling strong: ceil(N / world) rows per rank (configs[3] = 100M rows over 8 GPUs = 12.5M each)."""
    import bench
    from vq_amd import synth
    from vq_amd.dist import make_sharding

    for name, world, want in (("c4", 8, 12_500_000), ("c5", 8, 6_250_000), ("c4", 4, 25_000_000), ("c2", 3, 3_333_334)):
        cfg = synth.CONFIGS[name]
        rows = (cfg.N + world - 1) // world
        assert rows == want
        assert make_sharding(rows, world, 0, 1).total_rows >= cfg.N
    assert bench.parse_args(["--gpus", "8", "--scaling", "strong", "--workload", "c4"]).scaling == "strong"
