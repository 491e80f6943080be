"""Parity protocol between the device path (f32) and the f64 oracle:

  - winner index must equal the oracle's; a mismatch is tolerated only when the oracle's own
    winner / runner-up gap is a near-tie: (d2 - d1) / max(d1, tiny) < TIE_TAU.  Near-ties are
    counted and reported, never silently dropped.
  - winner distance within DIST_RTOL relative (plus an absolute floor for d -> 0), measured
    against the oracle distance of the prototype the device chose.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

TIE_TAU = 1e-5      # BASELINE.md "Parity bar"
DIST_RTOL = 1e-5    # BASELINE.json north_star: "distances must agree within 1e-5 relative"


@dataclass
class ParityReport:
    n: int
    mismatches: int          # device index != oracle index
    near_ties: int           # ... of which the oracle gap is below TIE_TAU (tolerated)
    hard: int                # ... of which it is not (failures)
    max_rel_err: float
    dist_fail: int
    floor_used: int = 0      # distances beyond rtol that pass only through the absolute floor `atol`

    @property
    def ok(self) -> bool:
        return self.hard == 0 and self.dist_fail == 0

    def __str__(self):
        return (f"n={self.n} mismatches={self.mismatches} near_ties={self.near_ties} hard={self.hard} "
                f"max_rel_err={self.max_rel_err:.3g} dist_fail={self.dist_fail} floor_used={self.floor_used}")


def compare(idx, dist, o_idx, o_best, o_second, rtol=DIST_RTOL, atol=0.0, tau=TIE_TAU) -> ParityReport:
    idx = np.asarray(idx).astype(np.int64)
    o_idx = np.asarray(o_idx).astype(np.int64)
    o_best = np.asarray(o_best, dtype=np.float64)
    o_second = np.asarray(o_second, dtype=np.float64)
    n = idx.size
    mism = idx != o_idx
    with np.errstate(invalid="ignore", divide="ignore"):
        gap = (o_second - o_best) / np.maximum(o_best, np.finfo(np.float64).tiny)
    near = mism & np.isfinite(gap) & (gap < tau)
    hard = mism & ~near
    max_rel, dist_fail, floor_used = 0.0, 0, 0
    if dist is not None:
        d = np.asarray(dist, dtype=np.float64)
        want = np.where(mism, o_second, o_best)           # a tolerated mismatch picked the runner-up
        fin = np.isfinite(want)
        # no eligible winner: the device must report +INF as well
        dist_fail += int((~fin & ~(np.isinf(d) & (d > 0))).sum())
        err = np.abs(d[fin] - want[fin])
        bad = err > rtol * np.abs(want[fin]) + atol
        bad &= ~hard[fin]                                  # do not double count hard index failures
        dist_fail += int(bad.sum())
        floor_used = int(((err > rtol * np.abs(want[fin])) & ~bad & ~hard[fin]).sum()) if atol else 0
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = err / np.maximum(np.abs(want[fin]), max(atol, 1e-300) if atol else 1e-300)
        rel = rel[np.abs(want[fin]) > atol] if atol else rel[np.abs(want[fin]) > 0]
        max_rel = float(rel.max()) if rel.size else 0.0
    return ParityReport(n, int(mism.sum()), int(near.sum()), int(hard.sum()), max_rel, dist_fail, floor_used)
