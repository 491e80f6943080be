"""CPU, world_size 2, gloo: the sharding + gather bookkeeping of the multi-GPU path
(vq_amd/dist.py).  The assignment itself is replaced by the known answer of the structured
synthetic data (row g -> prototype g mod K); the GPU kernels are covered by the -m gpu tests."""
import os
import socket

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
