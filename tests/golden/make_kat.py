#!/usr/bin/env python3
"""Writes tests/golden/kat.json: hand-derived known-answer vectors for the hot path.

The reference (Rust) cannot be built or run in this environment, so these vectors are NOT
outputs of the reference.  Each one is derived by hand from what the reference's own tests
assert (cited per case, paths under the reference root) plus the semantics of
src/helpers.rs:55-132 and src/lgmlvq/lgmlvq.rs:436-506; every expected number below is
written out literally (nothing is computed by the oracle or by the device code).

Data = inputs + expected outputs only; no reference source text is stored.
"""
import json
import math
import os

TOY = [[5.0, 2.0], [5.0, 3.0], [5.0, 4.0], [-5.0, 10.0], [-5.0, 11.0], [-5.0, 12.0]]  # tests/vq.rs:168-175
NAN, INF = float("nan"), float("inf")

cases = [
    dict(
        name="vq_toy_centres",
        cite="tests/vq.rs:164-202 (labels :192, prototype centres :200-201); src/vq/vq.rs:149-157",
        kind="euclid", data=TOY,
        prototypes=[[-5.0, 11.0], [5.0, 3.0]], proto_names=["0", "1"],
        expect_idx=[1, 1, 1, 0, 0, 0], expect_names=["1", "1", "1", "0", "0", "0"],
        expect_dist=[1.0, 0.0, 1.0, 1.0, 0.0, 1.0],
    ),
    dict(
        name="lvq_toy_centres",
        cite="tests/lvq.rs:283-334 (labels :324, centres :332-333); same labels asserted for GLVQ tests/glvq.rs:324",
        kind="euclid", data=TOY,
        prototypes=[[5.0, 3.0], [-5.0, 11.0]], proto_names=["C0", "C1"],
        expect_idx=[0, 0, 0, 1, 1, 1], expect_names=["C0", "C0", "C0", "C1", "C1", "C1"],
        expect_dist=[1.0, 0.0, 1.0, 1.0, 0.0, 1.0],
    ),
    dict(
        name="gmlvq_initial_identity_omega",
        cite="src/gmlvq/gmlvq.rs:160-161 with :112-122 (Omega = I/sqrt(n)); src/helpers.rs:35-45 is squared; labels tests/gmlvq.rs:324",
        kind="omega", data=TOY,
        prototypes=[[5.0, 3.0], [-5.0, 11.0]], proto_names=["C0", "C1"],
        omega=[[1.0 / math.sqrt(2.0), 0.0], [0.0, 1.0 / math.sqrt(2.0)]],
        expect_idx=[0, 0, 0, 1, 1, 1], expect_names=["C0", "C0", "C0", "C1", "C1", "C1"],
        expect_dist=[0.5, 0.0, 0.5, 0.5, 0.0, 0.5], dist_rtol=1e-15,
    ),
    dict(
        name="liramlvq_rank1_omega",
        cite="src/liramlvq/liramlvq.rs:165-171 (M x N Omega), :299; labels tests/liramlvq.rs:367. Omega=[[1,0]] keeps dim 0 only; "
             "last sample differs from both prototypes by 5 in dim 0 -> exact tie -> index 0 (src/helpers.rs:70)",
        kind="omega", data=TOY + [[0.0, 100.0]],
        prototypes=[[5.0, 3.0], [-5.0, 11.0]], proto_names=["C0", "C1"],
        omega=[[1.0, 0.0]],
        expect_idx=[0, 0, 0, 1, 1, 1, 0], expect_names=["C0", "C0", "C0", "C1", "C1", "C1", "C0"],
        expect_dist=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 25.0],
    ),
    dict(
        name="tie_identical_prototypes_lowest_index",
        cite="src/helpers.rs:58-59,70 (strict '<' keeps the first of equal distances)",
        kind="euclid", data=[[1.0, 1.0], [4.0, 5.0]],
        prototypes=[[9.0, 9.0], [4.0, 1.0], [4.0, 1.0], [4.0, 1.0]], proto_names=["a", "b", "c", "d"],
        expect_idx=[1, 1], expect_names=["b", "b"], expect_dist=[3.0, 4.0],
    ),
    dict(
        name="all_nan_or_inf_gives_index_0",
        cite="src/helpers.rs:58-59,70 (NaN and +INF distances never satisfy '<' against +INF)",
        kind="euclid", data=[[NAN, 0.0], [INF, 0.0], [1.0, 1.0]],
        prototypes=[[9.0, 9.0], [1.0, 1.0]], proto_names=["a", "b"],
        expect_idx=[0, 0, 1], expect_names=["a", "a", "b"], expect_dist=[INF, INF, 0.0],
    ),
    dict(
        name="nan_prototype_never_wins",
        cite="src/helpers.rs:70",
        kind="euclid", data=[[0.0, 0.0]],
        prototypes=[[NAN, 0.0], [3.0, 4.0], [INF, 0.0]], proto_names=["a", "b", "c"],
        expect_idx=[1], expect_names=["b"], expect_dist=[5.0],
    ),
    dict(
        name="lgmlvq_local_omega_is_honoured",
        cite="src/lgmlvq/lgmlvq.rs:445 (prototype j measured with omegas[j]); Omega_0 = I/sqrt(2) "
             "(:157-158), Omega_1 = 10*I makes the Euclid-nearest prototype 1 lose",
        kind="local", data=[[2.0, 0.0], [3.0, 0.0]],
        prototypes=[[0.0, 0.0], [3.0, 0.0]], proto_names=["C0", "C1"],
        omega=[[[1.0 / math.sqrt(2.0), 0.0], [0.0, 1.0 / math.sqrt(2.0)]], [[10.0, 0.0], [0.0, 10.0]]],
        expect_idx=[0, 1], expect_names=["C0", "C1"], expect_dist=[2.0, 0.0], dist_rtol=1e-15,
    ),
    dict(
        name="matched_and_non_matched",
        cite="src/helpers.rs:91-132; call pattern src/glvq/glvq.rs:171-176 (matching, then non-matching)",
        kind="euclid", data=[[5.0, 4.0], [-5.0, 12.0]],
        prototypes=[[5.0, 3.0], [-5.0, 11.0], [5.0, 10.0]], proto_names=["C0", "C1", "C0"],
        labels=["C0", "C1"],
        expect_idx_matching=[0, 1], expect_dist_matching=[1.0, 1.0],
        expect_idx_non_matching=[1, 2], expect_dist_non_matching=[math.sqrt(149.0), math.sqrt(104.0)],
    ),
    dict(
        name="matched_nothing_eligible_gives_0",
        cite="src/helpers.rs:98,113,122 (no prototype carries the label / every prototype carries it -> index stays 0)",
        kind="euclid", data=[[1.0, 1.0]],
        prototypes=[[7.0, 7.0], [1.0, 1.0]], proto_names=["A", "A"],
        labels=["B"],
        expect_idx_matching=[0], expect_dist_matching=[INF],
        expect_idx_non_matching=[1], expect_dist_non_matching=[0.0],
        also=dict(labels=["A"], expect_idx_matching=[1], expect_dist_matching=[0.0],
                  expect_idx_non_matching=[0], expect_dist_non_matching=[INF]),
    ),
    dict(
        name="lgmlvq_matched_local",
        cite="src/lgmlvq/lgmlvq.rs:468-506; call pattern :194-199",
        kind="local", data=[[2.0, 0.0]],
        prototypes=[[0.0, 0.0], [3.0, 0.0], [2.0, 1.0]], proto_names=["C0", "C1", "C1"],
        omega=[[[1.0, 0.0], [0.0, 1.0]], [[10.0, 0.0], [0.0, 10.0]], [[2.0, 0.0], [0.0, 2.0]]],
        labels=["C1"],
        expect_idx_matching=[2], expect_dist_matching=[4.0],
        expect_idx_non_matching=[0], expect_dist_non_matching=[4.0],
    ),
]


def _enc(o):
    """JSON has no NaN/Infinity: encode them as strings."""
    if isinstance(o, float):
        if math.isnan(o):
            return "nan"
        if math.isinf(o):
            return "inf" if o > 0 else "-inf"
        return o
    if isinstance(o, list):
        return [_enc(v) for v in o]
    if isinstance(o, dict):
        return {k: _enc(v) for k, v in o.items()}
    return o


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
    with open(out, "w") as f:
        json.dump(_enc(cases), f, indent=1)
    print("wrote", out, len(cases), "cases")
