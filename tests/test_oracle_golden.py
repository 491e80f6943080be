"""CPU: the oracle (oracle/vq_oracle.c) against the golden known-answer vectors.

The vectors are hand-derived from what the reference's own tests assert
(tests/golden/make_kat.py cites each one); the reference itself cannot be run here.
"""
import math

import numpy as np
import pytest

import oracle
from conftest import assert_dist_close, intern_names, load_kat

CASES = load_kat()


def _omega(case):
    return np.array(case["omega"], dtype=np.float64) if "omega" in case else None


@pytest.mark.parametrize("case", [c for c in CASES if "expect_idx" in c], ids=lambda c: c["name"])
@pytest.mark.parametrize("hoist", [False, True], ids=["faithful", "hoisted"])
@pytest.mark.parametrize("as_f32", [False, True], ids=["f64in", "f32in"])
def test_predict_kat(case, hoist, as_f32):
    X = np.array(case["data"], dtype=np.float32 if as_f32 else np.float64)
    W = np.array(case["prototypes"], dtype=np.float64)
    idx, best, _ = oracle.assign(X, W, omega=_omega(case), hoist=hoist)
    assert idx.tolist() == case["expect_idx"]
    names = [case["proto_names"][i] for i in idx]                    # src/vq/vq.rs:153-156
    assert names == case["expect_names"]
    assert_dist_close(best, case["expect_dist"], case.get("dist_rtol", 0.0), case["name"])
    # single-sample entry points agree with the batch driver
    for i, row in enumerate(X):
        assert oracle.find_closest(W, row, omega=_omega(case)) == case["expect_idx"][i]


def _check_matched(case, labels, exp):
    X = np.array(case["data"], dtype=np.float64)
    W = np.array(case["prototypes"], dtype=np.float64)
    table = intern_names(case["proto_names"])
    cls = np.array([table[n] for n in case["proto_names"]], dtype=np.uint32)
    lab = np.array([table.get(l, 0xFFFFFFFF) for l in labels], dtype=np.uint32)
    for matching, tag in ((True, "matching"), (False, "non_matching")):
        idx, best, _ = oracle.assign(X, W, omega=_omega(case), proto_cls=cls, labels=lab, matching=matching)
        assert idx.tolist() == exp[f"expect_idx_{tag}"], tag
        assert_dist_close(best, exp[f"expect_dist_{tag}"], 1e-15, tag)
        for i, row in enumerate(X):
            assert oracle.find_closest(W, row, omega=_omega(case), proto_cls=cls, label=lab[i],
                                       matching=matching) == exp[f"expect_idx_{tag}"][i]


@pytest.mark.parametrize("case", [c for c in CASES if "labels" in c], ids=lambda c: c["name"])
def test_matched_kat(case):
    _check_matched(case, case["labels"], case)
    if "also" in case:
        _check_matched(case, case["also"]["labels"], case["also"])


def test_distance_conventions():
    # Euclidean is rooted (helpers.rs:23), the Omega distance is squared (helpers.rs:44).
    assert oracle.euclidean_distance([0.0, 0.0], [3.0, 4.0]) == 5.0
    assert oracle.generalized_distance(np.eye(2), [0.0, 0.0], [3.0, 4.0]) == 25.0
    # argument order sample - prototype does not matter for the value, but the quadratic form must
    # use Lambda = Omega^T Omega, not Omega Omega^T: pick a non-symmetric rectangular Omega.
    om = np.array([[1.0, 2.0, 0.0]])
    v1, v2 = np.array([1.0, 1.0, 5.0]), np.array([0.0, 0.0, 0.0])
    assert oracle.generalized_distance(om, v1, v2) == 9.0       # (1*1 + 2*1 + 0*5)^2


def test_normalize_omega_trace_one():
    rng = np.random.default_rng(42)
    om = rng.uniform(-1, 1, size=(3, 5))
    n = oracle.normalize_omega(om)
    assert abs(np.trace(n.T @ n) - 1.0) < 1e-14                 # gmlvq.rs:108-121
    eye = oracle.normalize_omega(np.eye(4))
    assert np.allclose(eye, np.eye(4) / 2.0, rtol=0, atol=1e-16)  # gmlvq.rs:160-161: I/sqrt(n)


def test_faithful_equals_hoisted_and_threads():
    rng = np.random.default_rng(7)
    D, K, M, N = 6, 5, 4, 64
    X = rng.normal(size=(N, D)).astype(np.float32)
    W = rng.normal(size=(K, D))
    om = oracle.normalize_omega(rng.uniform(-1, 1, size=(M, D)))
    oms = np.stack([oracle.normalize_omega(rng.uniform(-1, 1, size=(M, D))) for _ in range(K)])
    for omega in (None, om, oms):
        a = oracle.assign(X, W, omega=omega, hoist=False, threads=1)
        b = oracle.assign(X, W, omega=omega, hoist=True, threads=3)
        assert (a[0] == b[0]).all()
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all()    # same operations -> same bits
    # against an independent numpy formulation ||Omega (x - w)||^2 (tolerance: different sum order)
    idx, best, second = oracle.assign(X, W, omega=om)
    ref = (((X.astype(np.float64)[:, None, :] - W[None]) @ om.T) ** 2).sum(-1)
    assert (idx == ref.argmin(1)).all()
    assert np.allclose(best, ref.min(1), rtol=1e-12)
    assert np.allclose(second, np.sort(ref, axis=1)[:, 1], rtol=1e-12)


def test_projection_restatements():
    rng = np.random.default_rng(3)
    om = rng.uniform(-1, 1, size=(2, 4))
    X = rng.normal(size=(5, 4))
    assert np.allclose(oracle.project_omega(om, X), X @ om.T, rtol=1e-14)
    assert np.allclose(oracle.project_lambda(om, X), X @ (om.T @ om).T, rtol=1e-13)
    assert math.isfinite(oracle.max_threads())
