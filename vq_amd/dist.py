"""Sample sharding across the GPUs of one node and the only exchange of the path: the gather of
winner indices and distances (RCCL over xGMI through torch.distributed, backend "nccl").

Every sample is independent (src/glvq/glvq.rs:224-232 carries no cross-sample state), so ranks
never exchange inputs: prototypes / class ids / Omega are replicated (<= 4 MiB), samples are
sharded, and results are all-gathered.

Layout: block-cyclic by chunk.  With W ranks, C chunks per rank and m rows per chunk, chunk c of
rank r holds global rows [(c*W + r)*m, (c*W + r + 1)*m).  The all-gather of chunk c therefore fills
the contiguous global range [c*W*m, (c+1)*W*m) in natural order, and can run on a side stream
while the next chunk's kernel is running.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Sharding:
    world: int
    rank: int
    chunks: int
    rows_per_chunk: int

    @property
    def rows_per_rank(self) -> int:
        return self.chunks * self.rows_per_chunk

    @property
    def total_rows(self) -> int:
        return self.world * self.rows_per_rank

    def first_row(self, chunk: int) -> int:
        """Global index of the first row of this rank's chunk."""
        return (chunk * self.world + self.rank) * self.rows_per_chunk

    def gathered_range(self, chunk: int):
        """Global row range [lo, hi) that the all-gather of `chunk` fills."""
        lo = chunk * self.world * self.rows_per_chunk
        return lo, lo + self.world * self.rows_per_chunk


def make_sharding(rows_per_rank: int, world: int, rank: int, chunks: int) -> Sharding:
    chunks = max(1, int(chunks))
    m = (rows_per_rank + chunks - 1) // chunks
    return Sharding(world, rank, chunks, m)


def gather_chunk(sh: Sharding, chunk: int, idx_all, dist_all, idx_local, dist_local, group=None):
    """all_gather of one chunk's results into their natural-order slice of the global arrays.
    Works on whatever backend the process group has (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist

    lo, hi = sh.gathered_range(chunk)
    dist.all_gather_into_tensor(idx_all[lo:hi], idx_local, group=group)
    if dist_all is not None:
        dist.all_gather_into_tensor(dist_all[lo:hi], dist_local, group=group)
