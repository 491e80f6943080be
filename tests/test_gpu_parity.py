"""GPU: parity of the HIP path (through the C ABI) with the f64 oracle.

Bar: winner indices identical to the oracle (near-ties, relative gap < 1e-5,
flagged and counted), winner distances within 1e-5 relative.  Euclidean distances rooted, Omega
distances squared (src/helpers.rs:23,44).  Tolerances are written out in tests/parity.py.
"""
import math

import numpy as np
import pytest

import oracle
import parity
from conftest import assert_dist_close, intern_names, load_kat
from vq_amd import KERNEL_AUTO, KERNEL_MFMA, KERNEL_SCALAR, Context, VqhipError, synth

pytestmark = pytest.mark.gpu

CASES = load_kat()
POLICIES = [KERNEL_SCALAR, KERNEL_AUTO]
POLICY_IDS = ["scalar", "auto"]


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _load_case(ctx, case, with_labels=False):
    W = np.array(case["prototypes"], dtype=np.float32)
    ctx.set_prototypes(W)
    table = intern_names(case["proto_names"])
    if with_labels:
        ctx.set_labels(np.array([table[n] for n in case["proto_names"]], dtype=np.uint32))
    if case["kind"] == "omega":
        ctx.set_omega(np.array(case["omega"], dtype=np.float32))
    elif case["kind"] == "local":
        ctx.set_local_omegas(np.array(case["omega"], dtype=np.float32))
    return table


# ------------------------------------------------------------------ golden vectors via the C ABI

@pytest.mark.parametrize("policy", POLICIES, ids=POLICY_IDS)
@pytest.mark.parametrize("case", [c for c in CASES if "expect_idx" in c], ids=lambda c: c["name"])
def test_kat_predict(ctx, case, policy):
    ctx.set_kernel_policy(policy)
    _load_case(ctx, case)
    X = np.array(case["data"], dtype=np.float32)
    idx, dist = ctx.assign(X)
    assert idx.tolist() == case["expect_idx"]
    assert [case["proto_names"][i] for i in idx] == case["expect_names"]       # src/vq/vq.rs:153-156
    assert_dist_close(dist, case["expect_dist"], 1e-6, case["name"])           # f32 rounding of Omega = I/sqrt(2)


def _matched(ctx, case, labels, exp, table):
    X = np.array(case["data"], dtype=np.float32)
    y = np.array([table.get(l, 0xFFFFFFFF) for l in labels], dtype=np.uint32)
    jp, dp, jn, dn = ctx.assign_matched(X, y)
    assert jp.tolist() == exp["expect_idx_matching"]
    assert jn.tolist() == exp["expect_idx_non_matching"]
    assert_dist_close(dp, exp["expect_dist_matching"], 1e-6)
    assert_dist_close(dn, exp["expect_dist_non_matching"], 1e-6)


@pytest.mark.parametrize("policy", POLICIES, ids=POLICY_IDS)
@pytest.mark.parametrize("case", [c for c in CASES if "labels" in c], ids=lambda c: c["name"])
def test_kat_matched(ctx, case, policy):
    ctx.set_kernel_policy(policy)
    table = _load_case(ctx, case, with_labels=True)
    _matched(ctx, case, case["labels"], case, table)
    if "also" in case:
        _matched(ctx, case, case["also"]["labels"], case["also"], table)


# ------------------------------------------------------------------ seeded models against the oracle

SHAPES = [
    # name, model, N, D, K, M, omega
    ("c1_vq_iris_shape", "vq::VQ", 150, 4, 3, 0, "none"),
    ("glvq_128x256", "GLVQ", 4099, 128, 256, 0, "none"),
    ("glvq_ragged_d100_k37", "GLVQ", 777, 100, 37, 0, "none"),
    ("glvq_d64_k70", "GLVQ", 1500, 64, 70, 0, "none"),
    ("glvq_streamed_k600", "GLVQ", 2500, 128, 600, 0, "none"),       # image larger than LDS: k_engine
    ("glvq_streamed_d256", "GLVQ", 1200, 256, 300, 0, "none"),
    ("gmlvq_128x128", "GMLVQ", 2051, 128, 256, 128, "global"),
    ("gmlvq_small_d12", "GMLVQ", 500, 12, 9, 12, "global"),
    ("liramlvq_r16_d256", "LiRaMLVQ", 1111, 256, 128, 16, "global"),
    ("liramlvq_r3_d20", "LiRaMLVQ", 300, 20, 6, 3, "global"),
    ("lgmlvq_8x128", "LGMLVQ", 333, 128, 8, 128, "local"),
    ("lgmlvq_small", "LGMLVQ", 257, 10, 5, 10, "local"),
    # BASELINE.json configs[3] / configs[4] at their own model shapes (rows cut to what the f64 oracle walks in seconds):
    # K = 1024 is the LDS-capacity edge of k_lowrank_pipe (88 KiB of projected prototypes + 16 KiB of Omega), 64 Omega_j
    # of 128 x 128 are the 4 MiB stream of k_engine<LOCAL> that does not fit L2 next to the rows
    ("c4_liramlvq_r16_d256_k1024", "LiRaMLVQ", 1024, 256, 1024, 16, "global"),
    ("c5_lgmlvq_64x128x128", "LGMLVQ", 1536, 128, 64, 128, "local"),
]


# kernel family VQHIP_KERNEL_AUTO must pick per shape (the MFMA kernels serve every D <= 256)
AUTO_KERNEL = {
    "c1_vq_iris_shape": "valu_euclid_direct", "glvq_128x256": "mfma_euclid", "glvq_ragged_d100_k37": "mfma_euclid",
    "glvq_d64_k70": "mfma_euclid", "glvq_streamed_k600": "mfma_euclid", "glvq_streamed_d256": "mfma_euclid", "gmlvq_128x128": "mfma_omega", "gmlvq_small_d12": "mfma_omega",
    "liramlvq_r16_d256": "mfma_omega", "liramlvq_r3_d20": "mfma_omega", "lgmlvq_8x128": "mfma_local_bf16",
    "lgmlvq_small": "mfma_local",  # D = 10: rows only 4-byte aligned, cut at D inside the kernel
    "c4_liramlvq_r16_d256_k1024": "mfma_omega_lowrank_lds", "c5_lgmlvq_64x128x128": "mfma_local_bf16",
}


def _expect_kernel(ctx, shape, policy):
    want = "scalar" if policy == KERNEL_SCALAR else AUTO_KERNEL[shape[0]]
    assert ctx.last_kernel.startswith(want), f"{shape[0]}: ran {ctx.last_kernel}, expected {want}"


def _model(shape):
    name, model, N, D, K, M, om = shape
    return synth.make_model(synth.Config(name, model, N, D, K, M=M, omega=om, classes=min(8, max(2, K // 2))))


def _atol(m):
    # Omega models subtract two f32 projections (Omega x - Omega w): absolute floor of a few ulps
    # of the projected magnitude, squared distance scale.
    return 0.0 if m.cfg.omega == "none" else 2e-6


# shapes of BASELINE.json configs 3 / 4 / 5: DESIGN.md section 2 says the absolute floor is never what lets them pass
NO_FLOOR = ("gmlvq_128", "c4_", "c5_", "liramlvq_r16")


@pytest.mark.parametrize("policy", POLICIES, ids=POLICY_IDS)
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: s[0])
def test_predict_parity(ctx, shape, policy):
    m = _model(shape)
    X = synth.samples_host(m, 0, m.cfg.N)
    ctx.set_kernel_policy(policy)
    synth.load_model(ctx, m, with_labels=False)
    idx, dist = ctx.assign(X)
    o_idx, o_best, o_second = oracle.assign(X, m.W, omega=m.omega, threads=8)
    rep = parity.compare(idx, dist, o_idx, o_best, o_second, atol=_atol(m))
    print(shape[0], ctx.last_kernel, rep)
    assert rep.ok, str(rep)
    if shape[0].startswith(NO_FLOOR):
        assert rep.floor_used == 0, f"{ctx.last_kernel}: {rep}"
    _expect_kernel(ctx, shape, policy)
    # structured data: sample i was drawn around prototype i mod K (not guaranteed for Omega models)
    if m.cfg.omega == "none" and m.cfg.D >= 32:
        assert (idx == np.arange(m.cfg.N) % m.cfg.K).all()


@pytest.mark.parametrize("policy", POLICIES, ids=POLICY_IDS)
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: s[0])
def test_matched_parity(ctx, shape, policy):
    m = _model(shape)
    N = min(m.cfg.N, 1024)
    X = synth.samples_host(m, 0, N)
    y = synth.sample_labels(m, 0, N)
    ctx.set_kernel_policy(policy)
    synth.load_model(ctx, m, with_labels=True)
    jp, dp, jn, dn = ctx.assign_matched(X, y)
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        o = oracle.assign(X, m.W, omega=m.omega, proto_cls=m.cls, labels=y, matching=matching, threads=8)
        rep = parity.compare(j, d, *o, atol=_atol(m))
        print(shape[0], "matching" if matching else "non-matching", ctx.last_kernel, rep)
        assert rep.ok, str(rep)
        if shape[0].startswith(NO_FLOOR):
            assert rep.floor_used == 0, f"{ctx.last_kernel} matching={matching}: {rep}"
    _expect_kernel(ctx, shape, policy)


def test_unstructured_data_parity(ctx):
    """Samples unrelated to the prototypes: small winner/runner-up gaps, the hard case for argmin."""
    rng = np.random.default_rng(42)
    D, K, N = 128, 256, 8192
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    o = oracle.assign(X, W, threads=8)
    for policy in POLICIES:
        ctx.set_kernel_policy(policy)
        ctx.set_prototypes(W)
        idx, dist = ctx.assign(X)
        rep = parity.compare(idx, dist, *o)
        print(ctx.last_kernel, rep)
        assert rep.ok, str(rep)


# ------------------------------------------------------------------ edge cases

def test_empty_and_single(ctx):
    ctx.set_kernel_policy(KERNEL_AUTO)
    W = np.array([[0.0, 0.0], [1.0, 1.0]], dtype=np.float32)
    ctx.set_prototypes(W)
    idx, dist = ctx.assign(np.zeros((0, 2), dtype=np.float32))
    assert idx.shape == (0,) and dist.shape == (0,)
    idx, dist = ctx.assign(np.array([[0.9, 0.9]], dtype=np.float32))
    assert idx.tolist() == [1] and abs(dist[0] - math.sqrt(0.02)) < 1e-7


def test_nan_and_inf_rows_inside_a_batch(ctx):
    m = _model(("t", "GLVQ", 700, 128, 256, 0, "none"))
    X = synth.samples_host(m, 0, 700).copy()
    X[5, 17] = np.nan
    X[300, 0] = np.inf
    X[699, 127] = -np.inf
    o = oracle.assign(X, m.W)
    assert o[0][5] == 0 and o[0][300] == 0 and o[0][699] == 0       # helpers.rs:58-59,70
    for policy in POLICIES:
        ctx.set_kernel_policy(policy)
        synth.load_model(ctx, m, with_labels=False)
        idx, dist = ctx.assign(X)
        rep = parity.compare(idx, dist, *o)
        assert rep.ok, str(rep)
        assert np.isinf(dist[[5, 300, 699]]).all()


def test_call_order_errors(ctx):
    c2 = Context(0)
    try:
        with pytest.raises(VqhipError) as e:
            c2.D = 2
            c2.assign(np.zeros((1, 2), dtype=np.float32))
        assert e.value.code == -2 and "prototypes" in e.value.text
        c2.set_prototypes(np.zeros((2, 2), dtype=np.float32))
        with pytest.raises(VqhipError) as e:
            c2.assign_matched(np.zeros((1, 2), dtype=np.float32), np.zeros(1, dtype=np.uint32))
        assert e.value.code == -2 and "labels" in e.value.text
        with pytest.raises(VqhipError):
            c2.set_omega(np.zeros((2, 3), dtype=np.float32))        # D mismatch
        with pytest.raises(ValueError):
            c2.assign(np.zeros((1, 3), dtype=np.float32))           # "Data must have the same dimensions"
    finally:
        c2.close()


# ------------------------------------------------------------------ device-resident path

def test_device_path_equals_host_path_and_synth_matches(ctx):
    import torch

    m = _model(("t", "GLVQ", 5000, 128, 256, 0, "none"))
    Xh = synth.samples_host(m, 1000, 5000)
    Xd = synth.samples_device(ctx, m, 1000, 5000)
    assert (Xd.cpu().numpy() == Xh).all()                           # generator: host bits == device bits
    for policy in POLICIES:
        ctx.set_kernel_policy(policy)
        synth.load_model(ctx, m)
        ih, dh = ctx.assign(Xh)
        idd, ddd = ctx.assign(Xd)
        torch.cuda.synchronize()
        assert (idd.cpu().numpy().view(np.uint32) == ih).all()
        assert (ddd.cpu().numpy() == dh).all()
        y = torch.from_numpy(synth.sample_labels(m, 1000, 5000).view(np.int32)).cuda()
        jp, dp, jn, dn = ctx.assign_matched(Xd, y)
        hp = ctx.assign_matched(Xh, synth.sample_labels(m, 1000, 5000))
        torch.cuda.synchronize()
        for a, b in zip((jp, dp, jn, dn), hp):
            assert (a.cpu().numpy().view(b.dtype) == b).all()


def test_host_path_multi_chunk(ctx):
    """More rows than one pinned staging buffer holds (D=4 -> 8M-row chunks): N spans 2+ chunks."""
    m = _model(("t", "vq::VQ", 1, 4, 3, 0, "none"))
    N = (128 << 20) // 16 * 2 + 12345
    X = synth.samples_host(m, 0, N)
    ctx.set_kernel_policy(KERNEL_AUTO)
    synth.load_model(ctx, m, with_labels=False)
    idx, dist = ctx.assign(X)
    assert (idx == np.arange(N) % 3).all()
    for lo in (0, N // 2 - 500, N - 1000):
        o = oracle.assign(X[lo:lo + 1000], m.W)
        rep = parity.compare(idx[lo:lo + 1000], dist[lo:lo + 1000], *o)
        assert rep.ok, str(rep)


def test_host_path_multi_chunk_shared_scratch_and_queues(ctx):
    """Chunks alternate between two streams: the scalar Omega path shares one projection scratch buffer
    between them, the MFMA kernels one set of batch counters.  Every chunk must come out whole."""
    m = _model(("t", "LiRaMLVQ", 1, 300, 6, 3, "global"))             # 111K-row chunks
    N = (128 << 20) // 1200 * 2 + 4321
    X = synth.samples_host(m, 0, N)
    ctx.set_kernel_policy(KERNEL_SCALAR)                               # the scalar Omega path: project into scratch, search
    synth.load_model(ctx, m, with_labels=False)
    idx, dist = ctx.assign(X)
    assert ctx.last_kernel == "scalar"
    ctx.set_kernel_policy(KERNEL_AUTO)
    for lo in (0, N // 3, N // 2 - 300, N - 2000):
        o = oracle.assign(X[lo:lo + 2000], m.W, omega=m.omega, threads=8)
        rep = parity.compare(idx[lo:lo + 2000], dist[lo:lo + 2000], *o, atol=2e-6)
        assert rep.ok, f"rows {lo}..: {rep}"
    m = _model(("t", "GMLVQ", 1, 300, 6, 200, "global"))              # two passes through the shared projected-sample scratch
    X = synth.samples_host(m, 0, N)
    synth.load_model(ctx, m, with_labels=False)
    idx, dist = ctx.assign(X)
    assert ctx.last_kernel == "mfma_omega_two_pass"
    for lo in (0, N // 3, N // 2 - 300, N - 2000):
        o = oracle.assign(X[lo:lo + 2000], m.W, omega=m.omega, threads=8)
        rep = parity.compare(idx[lo:lo + 2000], dist[lo:lo + 2000], *o, atol=2e-6)
        assert rep.ok, f"two-pass rows {lo}..: {rep}"
    m = _model(("t", "GLVQ", 1, 16, 40, 0, "none"))                   # D = 16: pipelined kernel, 2M-row chunks
    N = (128 << 20) // 64 * 3 + 99
    X = synth.samples_host(m, 0, N)
    synth.load_model(ctx, m, with_labels=False)
    idx, dist = ctx.assign(X)
    assert ctx.last_kernel == "mfma_euclid_lds_pipe"
    ref_idx, ref_dist = ctx.assign(X[: N // 7])                        # one chunk: no overlap possible
    assert (idx[: N // 7] == ref_idx).all() and (dist[: N // 7] == ref_dist).all()
    for lo in (N // 3 - 1000, 2 * (N // 3), N - 2000):
        o = oracle.assign(X[lo:lo + 2000], m.W, threads=8)
        rep = parity.compare(idx[lo:lo + 2000], dist[lo:lo + 2000], *o)
        assert rep.ok, f"rows {lo}..: {rep}"


# ------------------------------------------------------------------ ties at scale (packed-key screening)

@pytest.mark.parametrize("copies", [(64, 4), (50, 5)], ids=["4x64", "5x50"])
@pytest.mark.parametrize("sign", ["positive_scores", "negative_scores"])
def test_duplicate_prototypes_keep_the_lowest_index(ctx, sign, copies):
    """Every prototype appears several times (k, k+B, k+2B, ...): all their distances are bit-identical and
    helpers.rs:70 keeps the first.  Exercises the tie rules of the pipelined kernel at every level
    (inside a lane's tile, across tiles, across the lane pair; B = 50 puts the copies at different
    positions of their tiles, where the packed keys alone would not keep the first); samples far from /
    close to the origin make the raw screening scores ||w||^2 - 2 x.w positive / negative.  The labelled
    searches give the copies different classes, so the first ELIGIBLE copy must win (helpers.rs:113,122)."""
    B, reps = copies
    rng = np.random.default_rng(7)
    D, N = 128, 4096
    base = rng.uniform(-1, 1, size=(B, D)).astype(np.float32)
    W = np.concatenate([base] * reps, axis=0)
    K = W.shape[0]
    X = (base[rng.integers(0, B, size=N)] + 0.05 * rng.uniform(-1, 1, size=(N, D))).astype(np.float32)
    if sign == "positive_scores":
        X *= np.float32(0.01)
    o = oracle.assign(X, W, threads=8)
    assert (o[0] < B).all()
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    for policy in POLICIES:
        ctx.set_kernel_policy(policy)
        ctx.set_prototypes(W)
        idx, dist = ctx.assign(X)
        rep = parity.compare(idx, dist, *o)
        assert rep.ok and rep.mismatches == 0, f"{ctx.last_kernel}: {rep}"
        ctx.set_labels(cls)
        jp, dp, jn, dn = ctx.assign_matched(X, y)
        for matching, j, d in ((True, jp, dp), (False, jn, dn)):
            om = oracle.assign(X, W, proto_cls=cls, labels=y, matching=matching, threads=8)
            rep = parity.compare(j, d, *om)
            assert rep.ok and rep.mismatches == 0, f"{ctx.last_kernel} matching={matching}: {rep}"


@pytest.mark.parametrize("K", [1, 2, 31, 33, 64, 65, 255, 256])
def test_pipelined_kernel_prototype_counts(ctx, K):
    """K around the tile (32) and tile-pair (64) boundaries: pad rows must never win."""
    rng = np.random.default_rng(K)
    D, N = 128, 2000
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    idx, dist = ctx.assign(X)
    # at most 8 prototypes: the direct-form kernel; one tile of prototypes leaves room in LDS for the rows: they are
    # staged through it (k_resident_pipe STAGED)
    assert ctx.last_kernel == ("valu_euclid_direct" if K <= 8 else "mfma_euclid_lds_staged" if K <= 32 else "mfma_euclid_lds_pipe")
    rep = parity.compare(idx, dist, *oracle.assign(X, W, threads=8))
    assert rep.ok, str(rep)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    cls = (np.arange(K) % 3).astype(np.uint32)
    ctx.set_labels(cls)
    jp, dp, jn, dn = ctx.assign_matched(X, y)
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        rep = parity.compare(j, d, *oracle.assign(X, W, proto_cls=cls, labels=y, matching=matching, threads=8))
        assert rep.ok, f"K={K} matching={matching}: {rep}"


def test_huge_and_tiny_magnitudes(ctx):
    """Rows whose ||x||^2 overflows f32 (exact scan) next to ordinary and denormal-scale rows."""
    rng = np.random.default_rng(3)
    D, K, N = 128, 256, 512
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    X[::7] *= np.float32(1e30)
    X[3::7] *= np.float32(1e-30)
    X[5::7] *= np.float32(3e18)
    o = oracle.assign(X, W, threads=8)
    for policy in POLICIES:
        ctx.set_kernel_policy(policy)
        ctx.set_prototypes(W)
        idx, dist = ctx.assign(X)
        fin = np.isfinite(dist)           # f32 overflows where f64 does not: compare the finite rows only
        rep = parity.compare(idx[fin], dist[fin], o[0][fin], o[1][fin], o[2][fin])
        assert rep.ok, f"{ctx.last_kernel}: {rep}"
        assert fin[3::7].all() and fin[1::7].all()


# ------------------------------------------------------------------ f64 entry points

def test_f64_euclidean_path_is_bit_exact(ctx):
    """Same operations in the same order as src/helpers.rs:11-24, 55-77: indices AND distances equal
    the oracle's bit for bit, near-ties included."""
    rng = np.random.default_rng(11)
    for (N, D, K) in ((3000, 128, 256), (500, 7, 5), (64, 1, 2)):
        W = rng.uniform(-1, 1, size=(K, D))
        X = rng.uniform(-1, 1, size=(N, D))
        X[N // 2] = W[K - 1]                       # an exact hit: distance 0
        W[0] = W[K - 1]                            # and a tie for it: index 0 must win
        ctx.set_prototypes_f64(W)
        idx, dist = ctx.assign_f64(X)
        assert ctx.last_kernel == "scalar_f64"
        o_idx, o_best, _ = oracle.assign(X, W, threads=8)
        assert (idx == o_idx).all()
        assert (dist.view(np.uint64) == np.asarray(o_best).view(np.uint64)).all()
        assert idx[N // 2] == 0 and dist[N // 2] == 0.0
        cls = (np.arange(K) % 2).astype(np.uint32)
        y = rng.integers(0, 2, size=N).astype(np.uint32)
        ctx.set_labels(cls)
        jp, dp, jn, dn = ctx.assign_f64(X, y)
        for matching, j, d in ((True, jp, dp), (False, jn, dn)):
            oi, ob, _ = oracle.assign(X, W, proto_cls=cls, labels=y, matching=matching, threads=8)
            assert (j == oi).all()
            assert (d.view(np.uint64) == np.asarray(ob).view(np.uint64)).all()


def test_f64_path_golden_vectors(ctx, kat):
    for case in kat:
        if case["kind"] != "euclid" or "expect_idx" not in case:
            continue
        ctx.set_prototypes_f64(np.array(case["prototypes"], dtype=np.float64))
        idx, dist = ctx.assign_f64(np.array(case["data"], dtype=np.float64))
        assert idx.tolist() == case["expect_idx"], case["name"]
        assert_dist_close(dist, case["expect_dist"], 1e-15, case["name"])


def test_assign_rows_f64_equals_contiguous_f32(ctx):
    """The reference's container (separately allocated f64 rows) through the packing entry point."""
    m = _model(("t", "GLVQ", 9000, 128, 256, 0, "none"))
    X32 = synth.samples_host(m, 0, 9000)
    rows = [np.array(r, dtype=np.float64) for r in X32]          # one allocation per row
    ctx.set_kernel_policy(KERNEL_AUTO)
    synth.load_model(ctx, m, with_labels=False)
    i0, d0 = ctx.assign(X32)
    i1, d1 = ctx.assign_rows(rows)
    assert (i0 == i1).all() and (d0 == d1).all()
    i2, _ = ctx.assign_rows([], want_dist=False)
    assert i2.shape == (0,)
    with pytest.raises(ValueError):
        ctx.assign_rows([np.zeros(128), np.zeros(127)])          # ragged row (src/helpers.rs:14)


# ------------------------------------------------------------------ projections

def _rel_err(got, want):
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-30)     # per row: entries can cancel to ~0
    return float((np.abs(got - want) / scale).max())


def test_project_rows_f64_equals_contiguous_f32(ctx):
    """vqhip_project_rows_f64 (the reference's container in and out, src/gmlvq/gmlvq.rs:354-372): the rows of vqhip_project on
    the same values, widened; more rows than one pinned chunk holds, project and project_omega, ragged input refused."""
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    rng = np.random.default_rng(77)
    D, M, K = 96, 12, 40
    N = (128 << 20) // (D * 4) + 1234                                       # two chunks of the pinned pipeline
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    ctx.set_prototypes(W)
    ctx.set_omega(Om)
    rows = [np.array(r, dtype=np.float64) for r in X]                        # one allocation per row
    for which, width in ((PROJECT_LAMBDA, D), (PROJECT_OMEGA, M)):
        want = ctx.project(X, which)
        got = ctx.project_rows(rows, which)
        assert len(got) == N and all(g.shape == (width,) for g in got[:5])
        assert np.array_equal(np.stack(got).astype(np.float32), want), which
    assert ctx.project_rows([], PROJECT_OMEGA) == []
    with pytest.raises(ValueError):
        ctx.project_rows([np.zeros(D), np.zeros(D - 1)])                   # ragged row (src/helpers.rs:14)


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[6] == "global"], ids=lambda s: s[0])
def test_project_global(ctx, shape):
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    m = _model(shape)
    X = synth.samples_host(m, 0, m.cfg.N)
    ctx.set_kernel_policy(KERNEL_AUTO)
    synth.load_model(ctx, m, with_labels=False)
    Y = ctx.project(X, PROJECT_LAMBDA)                                    # gmlvq.rs:354-372
    assert Y.shape == X.shape
    assert _rel_err(Y, oracle.project_lambda(m.omega, X)) < 1e-5
    lam_kernel = ctx.last_kernel
    Z = ctx.project(X, PROJECT_OMEGA)                                     # liramlvq.rs:429-446
    assert Z.shape == (m.cfg.N, m.cfg.M)
    assert _rel_err(Z, oracle.project_omega(m.omega, X)) < 1e-5
    # the matrix-core kernel serves every D <= 256 and every output width
    assert lam_kernel == "mfma_project" and ctx.last_kernel == "mfma_project"
    ctx.set_kernel_policy(KERNEL_SCALAR)                                  # the scalar kernels stay the cross-check
    assert _rel_err(ctx.project(X, PROJECT_LAMBDA), oracle.project_lambda(m.omega, X)) < 1e-5
    assert ctx.last_kernel == "scalar_project"
    ctx.set_kernel_policy(KERNEL_AUTO)


@pytest.mark.parametrize("case", [(1, 128, 128), (33, 64, 20), (4097, 256, 16), (1000, 256, 200), (95, 32, 7), (64, 128, 36),
                                  (777, 126, 5), (300, 10, 10), (2049, 200, 33), (65, 3, 2)],
                         ids=lambda c: "n%d_d%d_m%d" % c)
def test_project_mfma_kernel(ctx, case):
    """k_project: row counts around the 32-row batch, output widths that end inside a tile / inside a group of
    8 columns / that are not multiples of 4 (4-byte stores), row lengths off the power-of-two grid (cut at D),
    Lambda of 256 x 256 (two column blocks), nothing written past N (gmlvq.rs:354-372, liramlvq.rs:401-446)."""
    import torch
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    N, D, M = case
    rng = np.random.default_rng(N + D + M)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(rng.uniform(-1, 1, size=(3, D)).astype(np.float32))
    ctx.set_omega(Om)
    Xd = torch.from_numpy(X).cuda()
    for which, width, want in ((PROJECT_LAMBDA, D, oracle.project_lambda(Om, X[:N])), (PROJECT_OMEGA, M, oracle.project_omega(Om, X[:N]))):
        Y = torch.full((N + 2, width), -7.0, dtype=torch.float32, device="cuda")
        ctx.project_into(Xd[:N], Y[:N], which)
        torch.cuda.synchronize()
        assert ctx.last_kernel == "mfma_project"
        assert (Y[N:].cpu().numpy() == -7.0).all()
        assert _rel_err(Y[:N].cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize("case", [(500, 784, 16), (300, 300, 40), (200, 512, 128), (65, 1001, 3), (1000, 257, 100), (33, 2049, 64)],
                         ids=lambda c: "n%d_d%d_m%d" % c)
def test_project_long_rows(ctx, case):
    """k_project_long: project / project_omega for a global Omega of at most 128 rows over rows above 256 values
    (gmlvq.rs:354-372, liramlvq.rs:401-446): Omega x over chunks of 64 values, then Omega^T y out of the accumulators;
    row lengths and output widths that are not multiples of 4 or 32, nothing written past N."""
    import torch
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    N, D, M = case
    rng = np.random.default_rng(N + D + M)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(rng.uniform(-1, 1, size=(3, D)).astype(np.float32))
    ctx.set_omega(Om)
    Xd = torch.from_numpy(X).cuda()
    for which, width, want in ((PROJECT_LAMBDA, D, oracle.project_lambda(Om, X[:N])), (PROJECT_OMEGA, M, oracle.project_omega(Om, X[:N]))):
        Y = torch.full((N + 2, width), -7.0, dtype=torch.float32, device="cuda")
        ctx.project_into(Xd[:N], Y[:N], which)
        torch.cuda.synchronize()
        assert ctx.last_kernel == "mfma_project_long"
        assert (Y[N:].cpu().numpy() == -7.0).all()
        assert _rel_err(Y[:N].cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[6] == "local"], ids=lambda s: s[0])
def test_project_local(ctx, shape):
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    m = _model(shape)
    X = synth.samples_host(m, 0, m.cfg.N)
    ctx.set_kernel_policy(KERNEL_AUTO)
    synth.load_model(ctx, m, with_labels=False)
    Y = ctx.project(X, PROJECT_LAMBDA)                                    # lgmlvq.rs:372-393
    win, best, second = oracle.assign(X, m.W, omega=m.omega, threads=8)
    om = np.asarray(m.omega, dtype=np.float64)
    want = np.stack([om[j].T @ (om[j] @ x.astype(np.float64)) for j, x in zip(win, X)])
    clear = (second - best) / np.maximum(best, 1e-300) > 1e-5              # near-tied winners may differ
    assert _rel_err(Y[clear], want[clear]) < 1e-5
    assert ctx.last_kernel == "mfma_project_local"
    ctx.set_kernel_policy(KERNEL_SCALAR)                                  # the scalar kernels stay the cross-check
    assert _rel_err(ctx.project(X, PROJECT_LAMBDA)[clear], want[clear]) < 1e-5
    assert ctx.last_kernel == "scalar_project"
    ctx.set_kernel_policy(KERNEL_AUTO)
    with pytest.raises(VqhipError) as e:
        ctx.project(X, PROJECT_OMEGA)
    assert e.value.code == -5
    ctx.set_prototypes(m.W)                                               # Euclidean model: nothing to project with
    with pytest.raises(VqhipError) as e:
        ctx.project(X, PROJECT_LAMBDA)
    assert e.value.code == -2


@pytest.mark.parametrize("case", [(1, 128, 16, 4), (33, 30, 30, 3), (1000, 100, 20, 7), (4097, 128, 16, 64), (500, 250, 8, 5),
                                  (65, 3, 2, 2), (3000, 64, 64, 1), (2500, 32, 5, 300)],
                         ids=lambda c: "n%d_d%d_m%d_k%d" % c)
def test_project_local_grouped_kernel(ctx, case):
    """LGMLVQ::project on the matrix cores (lgmlvq.rs:372-393): samples grouped by winner on the device, every group
    padded to whole batches of 32; groups of every size (empty ones, one that holds everything, more prototypes than
    a batch has rows), row lengths off the grid, nothing written past N."""
    import torch
    from vq_amd import PROJECT_LAMBDA
    N, D, M, K = case
    rng = np.random.default_rng(N + D + M + K)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Oms = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    ctx.set_local_omegas(Oms)
    Xd = torch.from_numpy(X).cuda()
    Y = torch.full((N + 2, D), -7.0, dtype=torch.float32, device="cuda")
    for _ in range(2):                                                    # twice: the scratch is reused
        ctx.project_into(Xd[:N], Y[:N], PROJECT_LAMBDA)
        torch.cuda.synchronize()
    assert ctx.last_kernel == "mfma_project_local"
    assert (Y[N:].cpu().numpy() == -7.0).all()
    win, best, second = oracle.assign(X[:N], W, omega=Oms, threads=8)
    om = Oms.astype(np.float64)
    want = np.stack([om[j].T @ (om[j] @ x.astype(np.float64)) for j, x in zip(win, X[:N])])
    clear = (second - best) / np.maximum(best, 1e-300) > 1e-5
    assert clear.mean() > 0.9
    assert _rel_err(Y[:N].cpu().numpy()[clear], want[clear]) < 1e-5


@pytest.mark.parametrize("K", [256, 20], ids=["pipe_k256", "staged_k20"])
@pytest.mark.parametrize("N", [1, 31, 32, 33, 63, 65, 4097])
def test_pipelined_kernel_row_counts(ctx, N, K):
    """Batches of 32 rows claimed in blocks of 2: rows past N are dropped by the buffer range check
    (fragment loads, LDS-DMA pieces of the staged variant, and the stores)."""
    rng = np.random.default_rng(N)
    D = 128
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 3, D)).astype(np.float32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    import torch
    Xd = torch.from_numpy(X).cuda()
    idx = torch.full((N + 3,), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((N + 3,), -7.0, dtype=torch.float32, device="cuda")
    ctx.assign_into(Xd[:N], idx[:N], dist[:N])
    torch.cuda.synchronize()
    assert ctx.last_kernel == ("mfma_euclid_lds_pipe" if K == 256 else "mfma_euclid_lds_staged")
    assert (idx[N:].cpu().numpy() == -7).all() and (dist[N:].cpu().numpy() == -7.0).all()   # nothing written past N
    rep = parity.compare(idx[:N].cpu().numpy(), dist[:N].cpu().numpy(), *oracle.assign(X[:N], W, threads=8))
    assert rep.ok, str(rep)


# ------------------------------------------------------------------ resident low-rank Omega kernel (LiRaMLVQ shapes)

LOWRANK = [(1, 256, 7, 16), (33, 256, 33, 16), (4097, 64, 1000, 16), (700, 128, 65, 5), (2000, 32, 12, 1),
           # Omega about square: prototypes resident, Omega streamed from L2 (k_omega_pipe)
           (4097, 128, 256, 128), (65, 128, 1, 100), (900, 64, 70, 50), (31, 32, 10, 20),
           # row lengths off the power-of-two grid: the CUT variants of both kernels
           (700, 200, 300, 16), (1000, 50, 64, 7), (500, 100, 256, 100), (333, 50, 40, 30), (2049, 126, 33, 126)]


@pytest.mark.parametrize("case", LOWRANK, ids=lambda c: "n%d_d%d_k%d_m%d" % c)
def test_lowrank_resident_kernel(ctx, case):
    """Omega models with the projected prototypes resident in LDS (k_lowrank_pipe, k_omega_pipe): unstructured
    data, row counts around the 32-row batch, K around the tile boundaries, nothing written past N,
    plain and matched searches (helpers.rs:35-45, 55-77, 91-132)."""
    import torch
    N, D, K, M = case
    rng = np.random.default_rng(N + K)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 3, D)).astype(np.float32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    ctx.set_omega(Om)
    Xd = torch.from_numpy(X).cuda()
    idx = torch.full((N + 3,), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((N + 3,), -7.0, dtype=torch.float32, device="cuda")
    ctx.assign_into(Xd[:N], idx[:N], dist[:N])
    torch.cuda.synchronize()
    want_kernel = "mfma_omega_lowrank_lds" if M <= 16 else "mfma_omega_lds"
    assert ctx.last_kernel == want_kernel
    assert (idx[N:].cpu().numpy() == -7).all() and (dist[N:].cpu().numpy() == -7.0).all()
    rep = parity.compare(idx[:N].cpu().numpy(), dist[:N].cpu().numpy(), *oracle.assign(X[:N], W, omega=Om, threads=8), atol=2e-6)
    print(case, rep)
    assert rep.ok, str(rep)
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    ctx.set_labels(cls)
    jp, dp, jn, dn = ctx.assign_matched(X[:N], y)
    assert ctx.last_kernel == want_kernel
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        o = oracle.assign(X[:N], W, omega=Om, proto_cls=cls, labels=y, matching=matching, threads=8)
        rep = parity.compare(j, d, *o, atol=2e-6)
        assert rep.ok, f"{case} matching={matching}: {rep}"


def test_lowrank_ties_nan_and_streamed_fallback(ctx):
    rng = np.random.default_rng(5)
    D, M, N = 128, 16, 1500
    base = rng.uniform(-1, 1, size=(50, D)).astype(np.float32)
    W = np.concatenate([base, base, base], axis=0)                    # every prototype three times: first one wins
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = (base[rng.integers(0, 50, size=N)] + 0.05 * rng.uniform(-1, 1, size=(N, D))).astype(np.float32)
    X[7, 3] = np.nan
    X[40, 100] = np.inf
    X[1499, 0] = -np.inf
    o = oracle.assign(X, W, omega=Om, threads=8)
    assert (o[0] < 50).all() and o[0][7] == 0 and o[0][40] == 0 and o[0][1499] == 0
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    ctx.set_omega(Om)
    for Om_, want in ((Om, "mfma_omega_lowrank_lds"), (np.concatenate([Om] * 5, axis=0), "mfma_omega_lds")):
        o = oracle.assign(X, W, omega=Om_, threads=8)
        ctx.set_prototypes(W)
        ctx.set_omega(Om_)
        idx, dist = ctx.assign(X)
        assert ctx.last_kernel == want
        rep = parity.compare(idx, dist, *o, atol=2e-6)
        assert rep.ok and rep.mismatches == 0, f"{want}: {rep}"
        assert np.isinf(dist[[7, 40, 1499]]).all()
    # more projected prototypes than LDS holds: the streamed engine serves the same model family
    K2, D2, N2 = 2000, 256, 150
    W2 = rng.uniform(-1, 1, size=(K2, D2)).astype(np.float32)
    Om2 = (rng.uniform(-1, 1, size=(M, D2)) / np.sqrt(M * D2 / 3.0)).astype(np.float32)
    X2 = rng.uniform(-1, 1, size=(N2, D2)).astype(np.float32)
    ctx.set_prototypes(W2)
    ctx.set_omega(Om2)
    idx, dist = ctx.assign(X2)
    assert ctx.last_kernel == "mfma_omega"
    rep = parity.compare(idx, dist, *oracle.assign(X2, W2, omega=Om2, threads=8), atol=2e-6)
    assert rep.ok, str(rep)


# ------------------------------------------------------------------ seeded sweep over shapes (kernel selection edges)

def _sweep_shapes():
    rng = np.random.default_rng(2024)
    out = []
    for i in range(24):
        kind = ("none", "global", "local")[i % 3]
        D = int(rng.choice([4, 8, 12, 16, 20, 32, 48, 64, 100, 128, 192, 256]))
        K = int(rng.choice([1, 2, 5, 31, 32, 33, 64, 100, 257, 300])) if kind != "local" else int(rng.choice([1, 2, 5, 9]))
        M = 0 if kind == "none" else int(rng.choice([1, 4, 7, 16, 17, 32, D])) if kind == "global" else int(rng.choice([3, 8, min(D, 32)]))
        M = min(M, 256)
        N = int(rng.choice([1, 31, 65, 257, 700]))
        out.append((N, D, K, M, kind))
    return out


@pytest.mark.parametrize("shape", _sweep_shapes(), ids=lambda s: "n%d_d%d_k%d_m%d_%s" % s)
def test_shape_sweep_against_oracle(ctx, shape):
    """Unstructured data over a seeded sweep of (N, D, K, M, model kind): whatever kernel VQHIP_KERNEL_AUTO picks
    must meet the parity bar, plain and matched, and agree with the scalar kernels' choice of winners wherever
    the oracle's gap is clear."""
    N, D, K, M, kind = shape
    rng = np.random.default_rng(N * 7 + D * 3 + K + M)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    Om = None
    if kind == "global":
        Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    elif kind == "local":
        Om = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    atol = 0.0 if kind == "none" else 2e-6
    o_plain = oracle.assign(X, W, omega=Om, threads=8)
    for policy in POLICIES:
        ctx.set_kernel_policy(policy)
        ctx.set_prototypes(W)
        if kind == "global":
            ctx.set_omega(Om)
        elif kind == "local":
            ctx.set_local_omegas(Om)
        idx, dist = ctx.assign(X)
        rep = parity.compare(idx, dist, *o_plain, atol=atol)
        assert rep.ok, f"{shape} {ctx.last_kernel}: {rep}"
        ctx.set_labels(cls)
        jp, dp, jn, dn = ctx.assign_matched(X, y)
        for matching, j, d in ((True, jp, dp), (False, jn, dn)):
            o = oracle.assign(X, W, omega=Om, proto_cls=cls, labels=y, matching=matching, threads=8)
            rep = parity.compare(j, d, *o, atol=atol)
            assert rep.ok, f"{shape} {ctx.last_kernel} matching={matching}: {rep}"


def test_global_omega_dispatch_grid(ctx):
    """Every (D, M, K) on a grid around the kernels' boundaries (D 32..300, Omega rows 1..300) is served by a
    matrix-core kernel under VQHIP_KERNEL_AUTO and agrees with the scalar kernels on the device: same winner wherever
    the scalar runner-up is not within 1e-4, distances within 1e-5 relative (+2e-6)."""
    import torch
    rng = np.random.default_rng(2024)
    N = 160
    seen = set()
    for D in (32, 100, 128, 129, 200, 256, 257, 300):
        X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
        Xd = torch.from_numpy(X).cuda()
        for M in (1, 16, 17, 64, 65, 128, 129, 256, 257, 300):
            Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
            for K in (2, 40, 257):
                W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
                out = {}
                for policy in (KERNEL_SCALAR, KERNEL_AUTO):
                    ctx.set_kernel_policy(policy)
                    ctx.set_prototypes(W)
                    ctx.set_omega(Om)
                    idx = torch.empty(N, dtype=torch.int32, device="cuda")
                    dist = torch.empty(N, dtype=torch.float32, device="cuda")
                    ctx.assign_into(Xd, idx, dist)
                    torch.cuda.synchronize()
                    out[policy] = (idx.cpu().numpy(), dist.cpu().numpy().astype(np.float64), ctx.last_kernel)
                si, sd, sk = out[KERNEL_SCALAR]
                ai, ad, ak = out[KERNEL_AUTO]
                # (dozens of prototypes projected onto a line included: crowded samples take the kernels' exact scan)
                assert sk == "scalar" and ak.startswith("mfma_omega"), (D, M, K, sk, ak)
                seen.add(ak)
                # distances first: a different winner within tolerance shows up as a near-equal distance
                assert np.all(np.abs(ad - sd) <= 1e-4 * np.abs(sd) + 2e-6), (D, M, K, ak, float(np.abs(ad - sd).max()))
                same = ai == si
                assert same.mean() > 0.97, (D, M, K, ak, float(same.mean()))
    ctx.set_kernel_policy(KERNEL_AUTO)
    assert {"mfma_omega_lds", "mfma_omega_lowrank_lds", "mfma_omega", "mfma_omega_long", "mfma_omega_two_pass"} <= seen, seen


def _auto_vs_scalar(ctx, Xd, N, load, what, prefix, rtol=1e-4, atol=2e-6):
    """One model under both policies on the device; returns the AUTO kernel's name."""
    import torch
    out = {}
    for policy in (KERNEL_SCALAR, KERNEL_AUTO):
        ctx.set_kernel_policy(policy)
        load()
        idx = torch.empty(N, dtype=torch.int32, device="cuda")
        dist = torch.empty(N, dtype=torch.float32, device="cuda")
        ctx.assign_into(Xd, idx, dist)
        torch.cuda.synchronize()
        out[policy] = (idx.cpu().numpy(), dist.cpu().numpy().astype(np.float64), ctx.last_kernel)
    si, sd, sk = out[KERNEL_SCALAR]
    ai, ad, ak = out[KERNEL_AUTO]
    assert sk == "scalar" and ak.startswith(prefix), (what, sk, ak)
    assert np.all(np.abs(ad - sd) <= rtol * np.abs(sd) + atol), (what, ak, float(np.abs(ad - sd).max()))
    assert (ai == si).mean() > 0.97, (what, ak, float((ai == si).mean()))
    return ak


def test_euclid_and_local_dispatch_grids(ctx):
    """As test_global_omega_dispatch_grid, for Euclidean models (D 1..1000, K 1..5000) and local Omegas (D 16..400,
    1..130 rows each): no shape on the grids falls back to the scalar kernels, and both agree on the device."""
    import torch
    rng = np.random.default_rng(77)
    N = 160
    seen = set()
    for D in (1, 3, 16, 31, 64, 100, 128, 200, 256, 257, 500, 1000):
        X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
        Xd = torch.from_numpy(X).cuda()
        for K in (1, 2, 33, 256, 257, 1000, 5000):
            W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
            # (thousands of random prototypes on a line included: crowded samples take the kernels' exact scan)
            ak = _auto_vs_scalar(ctx, Xd, N, lambda: ctx.set_prototypes(W), (D, K), ("mfma_euclid", "valu_euclid"), rtol=1e-5, atol=1e-7)
            seen.add(ak)
    for D in (16, 100, 128, 200, 256, 300, 400):
        X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
        Xd = torch.from_numpy(X).cuda()
        for M in (1, 32, 33, 130):
            for K in (1, 3, 20):
                W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
                Oms = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)

                def load():
                    ctx.set_prototypes(W)
                    ctx.set_local_omegas(Oms)
                seen.add(_auto_vs_scalar(ctx, Xd, N, load, (D, M, K), "mfma_local"))
    ctx.set_kernel_policy(KERNEL_AUTO)
    assert {"valu_euclid_direct", "mfma_euclid", "mfma_euclid_lds", "mfma_euclid_lds_pipe", "mfma_euclid_long", "mfma_local_bf16", "mfma_local_long"} <= seen, seen


def test_matched_dispatch_grid(ctx):
    """The labelled search (rows a4 / a6) under VQHIP_KERNEL_AUTO against the scalar kernels on the device, over shapes
    of all three model kinds: both winners of every sample, every kernel family."""
    rng = np.random.default_rng(11)
    N, kernels = 300, set()
    cases = [(D, K, 0, "none") for D in (3, 16, 64, 100, 128, 200, 256, 300, 784) for K in (1, 2, 5, 33, 257, 1000)]
    cases += [(D, K, M, "global") for D in (32, 100, 128, 200, 256, 300) for M in (2, 16, 40, 128, 200) for K in (3, 40, 300)]
    cases += [(D, K, M, "local") for D in (16, 128, 200, 300) for M in (1, 33) for K in (2, 20)]
    for D, K, M, kind in cases:
        W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
        X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
        cls = (np.arange(K) % 3).astype(np.uint32)
        y = rng.integers(0, 3, size=N).astype(np.uint32)
        shape = (M, D) if kind == "global" else (K, M, D)
        Om = None if kind == "none" else (rng.uniform(-1, 1, size=shape) / np.sqrt(M * D / 3.0)).astype(np.float32)
        out = {}
        for policy in (KERNEL_SCALAR, KERNEL_AUTO):
            ctx.set_kernel_policy(policy)
            ctx.set_prototypes(W)
            if kind == "global":
                ctx.set_omega(Om)
            elif kind == "local":
                ctx.set_local_omegas(Om)
            ctx.set_labels(cls)
            out[policy] = ctx.assign_matched(X, y) + (ctx.last_kernel,)
        s, a = out[KERNEL_SCALAR], out[KERNEL_AUTO]
        assert s[4] == "scalar" and a[4].startswith(("mfma_", "valu_")), ((D, K, M, kind), s[4], a[4])
        kernels.add(a[4])
        for si, sd, ai, ad in ((s[0], s[1], a[0], a[1]), (s[2], s[3], a[2], a[3])):
            sd, ad = sd.astype(np.float64), ad.astype(np.float64)
            fin = np.isfinite(sd)
            assert (np.isfinite(ad) == fin).all(), (D, K, M, kind, a[4])          # no eligible prototype: +INF on both
            assert np.all(np.abs(ad[fin] - sd[fin]) <= 1e-4 * np.abs(sd[fin]) + 2e-6), (D, K, M, kind, a[4])
            assert (si == ai).mean() > 0.97, (D, K, M, kind, a[4])
    ctx.set_kernel_policy(KERNEL_AUTO)
    assert len(kernels) >= 12, kernels


@pytest.mark.parametrize("case", [(5000, 128, 32), (3001, 64, 64), (2500, 64, 7), (1999, 32, 50)], ids=lambda c: "n%d_d%d_k%d" % c)
def test_staged_kernel(ctx, case):
    """Few prototypes: rows staged through LDS by 1 KiB LDS-DMA pieces (2 / 4 / 8 rows per piece at D = 128 / 64 / 32).
    Plain and matched search, NaN / INF rows, against the oracle and against the fragment-load kernel bit for bit."""
    import os
    N, D, K = case
    rng = np.random.default_rng(N + D + K)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    X[3, 1] = np.nan
    X[N - 1, D - 1] = np.inf
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    idx, dist = ctx.assign(X)
    assert ctx.last_kernel == ("mfma_euclid_lds_staged" if K > 8 else "valu_euclid_direct")
    o = oracle.assign(X, W, threads=8)
    rep = parity.compare(idx, dist, *o)
    assert rep.ok, str(rep)
    assert idx[3] == 0 and np.isinf(dist[3]) and idx[N - 1] == 0 and np.isinf(dist[N - 1])
    ctx.set_labels(cls)
    got = ctx.assign_matched(X, y)
    for matching, j, d in ((True, got[0], got[1]), (False, got[2], got[3])):
        rep = parity.compare(j, d, *oracle.assign(X, W, proto_cls=cls, labels=y, matching=matching, threads=8))
        assert rep.ok, f"matching={matching}: {rep}"
    os.environ["VQHIP_NO_STAGED"] = "1"                      # same model on the fragment-load kernel
    os.environ["VQHIP_NO_DIRECT"] = "1"
    try:
        ctx.set_prototypes(W)
        idx2, dist2 = ctx.assign(X)
        assert ctx.last_kernel == "mfma_euclid_lds_pipe"
        del os.environ["VQHIP_NO_STAGED"]
        ctx.set_prototypes(W)
        idx3, dist3 = ctx.assign(X)
        assert ctx.last_kernel == "mfma_euclid_lds_staged"
    finally:
        os.environ.pop("VQHIP_NO_STAGED", None)
        del os.environ["VQHIP_NO_DIRECT"]
    assert (idx2 == idx3).all() and (dist2.view(np.uint32) == dist3.view(np.uint32)).all()     # staged == fragment loads, bit for bit
    rep2 = parity.compare(idx2, dist2, *o)
    assert rep2.ok and (idx2 != idx).sum() <= rep.near_ties + rep2.near_ties, f"{rep} / {rep2}"


@pytest.mark.parametrize("D", [1, 2, 3, 5, 10, 30, 63, 101, 126, 255])
def test_row_lengths_that_are_not_multiples_of_four(ctx, D):
    """Rows of any length up to 256 run on the MFMA kernels: 16-byte fetches from 4-byte aligned rows, cut at D
    (the values behind a row are the next row's -- here NaN / INF poisoned on purpose -- or the end of the buffer)."""
    rng = np.random.default_rng(D)
    N, K = 3001, 37
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    X[1::2, 0] = np.nan                        # every second row starts with a NaN: its neighbours must not see it
    X[4::8, :] = np.inf
    clean = np.isfinite(X).all(axis=1)
    o = oracle.assign(X, W, threads=8)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    idx, dist = ctx.assign(X)
    # up to 32 values k_resident, above that the pipelined kernel's CUT variant (one register set above 128 values)
    want = "mfma_euclid_lds" if D <= 32 else "mfma_euclid_lds_pipe_cut"
    assert ctx.last_kernel == want, ctx.last_kernel
    rep = parity.compare(idx, dist, *o)
    assert rep.ok, str(rep)
    assert np.isfinite(dist[clean]).all() and (idx[~clean] == 0).all() and np.isinf(dist[~clean]).all()
    # the same model on the unpipelined padded-row kernel (k_resident): same winners wherever the gap is clear
    import os
    os.environ["VQHIP_NO_CUT"] = "1"
    try:
        ctx.set_prototypes(W)
        idx2, dist2 = ctx.assign(X)
        assert ctx.last_kernel == ("mfma_euclid_lds" if D <= 128 else "mfma_euclid")
    finally:
        del os.environ["VQHIP_NO_CUT"]
    rep2 = parity.compare(idx2, dist2, *o)
    assert rep2.ok and (idx2 != idx).sum() <= rep.near_ties + rep2.near_ties, f"{rep} / {rep2}"
    ctx.set_prototypes(W)
    if D >= 3:
        M = 3
        Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
        Xc = X[clean]
        ctx.set_omega(Om)
        idx, dist = ctx.assign(Xc)
        assert ctx.last_kernel.startswith("mfma_omega"), ctx.last_kernel
        rep = parity.compare(idx, dist, *oracle.assign(Xc, W, omega=Om, threads=8), atol=2e-6)
        assert rep.ok, str(rep)


@pytest.mark.parametrize("case", [(2000, 300, 50), (1500, 784, 10), (700, 257, 128), (333, 1000, 33), (65, 2049, 3), (4097, 512, 64),
                                  (500, 300, 200), (300, 513, 256),
                                  # more than 256 prototypes: groups of 8 tiles, the rows fetched again per group
                                  (400, 300, 257), (300, 784, 1000), (200, 400, 600)],
                         ids=lambda c: "n%d_d%d_k%d" % c)
def test_long_rows(ctx, case):
    """Rows above 256 values (k_long: the reduction in chunks of 128, candidates re-measured in a second sweep):
    unstructured data, plain and matched, NaN / INF / huge rows inside a batch, nothing written past N."""
    import torch
    N, D, K = case
    rng = np.random.default_rng(N + D + K)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    X[5 % N, D - 1] = np.nan
    X[40 % N, 0] = np.inf
    X[33 % N] *= np.float32(1e30)                       # ||x||^2 overflows f32: exact scan
    o = oracle.assign(X[:N], W, threads=8)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    Xd = torch.from_numpy(X).cuda()
    idx = torch.full((N + 2,), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((N + 2,), -7.0, dtype=torch.float32, device="cuda")
    ctx.assign_into(Xd[:N], idx[:N], dist[:N])
    torch.cuda.synchronize()
    assert ctx.last_kernel == "mfma_euclid_long"
    assert (idx[N:].cpu().numpy() == -7).all() and (dist[N:].cpu().numpy() == -7.0).all()
    gi, gd = idx[:N].cpu().numpy(), dist[:N].cpu().numpy()
    fin = np.isfinite(gd)                                # f32 overflows where f64 does not
    rep = parity.compare(gi[fin], gd[fin], o[0][fin], o[1][fin], o[2][fin])
    print(case, rep)
    assert rep.ok, str(rep)
    assert not fin[5 % N] and gi[5 % N] == 0 and not fin[40 % N] and gi[40 % N] == 0
    assert fin.sum() >= N - 3
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    ctx.set_labels(cls)
    jp, dp, jn, dn = ctx.assign_matched(X[:N], y)
    assert ctx.last_kernel == "mfma_euclid_long"
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        om = oracle.assign(X[:N], W, proto_cls=cls, labels=y, matching=matching, threads=8)
        f2 = np.isfinite(d) | ~np.isfinite(om[1])        # rows with no eligible prototype: +INF on both sides
        f2 &= fin | ~np.isfinite(om[1])
        rep = parity.compare(j[f2], d[f2], om[0][f2], om[1][f2], om[2][f2])
        assert rep.ok, f"{case} matching={matching}: {rep}"


@pytest.mark.parametrize("case", [(300, 300, 20, 300), (200, 784, 10, 200), (150, 400, 300, 129), (100, 512, 5, 513), (65, 260, 3, 260),
                                  # rows of at most 256 values whose Omega no single-pass kernel holds
                                  (300, 200, 20, 200), (200, 150, 300, 150), (100, 64, 5, 300), (150, 256, 10, 128)],
                         ids=lambda c: "n%d_d%d_k%d_m%d" % c)
def test_two_pass_omega(ctx, case):
    """A global Omega no single-pass kernel holds (GMLVQ's square Omega on rows of 129..256 or above 256 values, more
    than 256 Omega rows): two passes -- k_project_long (blocks of 128 Omega rows) or k_project into scratch, a
    Euclidean engine over W Omega^T in the projected space, distances squared -- against the oracle, plain and
    matched, NaN / INF rows, project_omega."""
    import torch
    from vq_amd import PROJECT_LAMBDA, PROJECT_OMEGA
    N, D, K, M = case
    rng = np.random.default_rng(N + D + K + M)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    X[5 % N, D - 1] = np.nan
    X[40 % N, 0] = np.inf
    o = oracle.assign(X[:N], W, omega=Om, threads=8)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    ctx.set_omega(Om)
    Xd = torch.from_numpy(X).cuda()
    idx = torch.full((N + 2,), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((N + 2,), -7.0, dtype=torch.float32, device="cuda")
    ctx.assign_into(Xd[:N], idx[:N], dist[:N])
    torch.cuda.synchronize()
    assert ctx.last_kernel == "mfma_omega_two_pass"
    assert (idx[N:].cpu().numpy() == -7).all() and (dist[N:].cpu().numpy() == -7.0).all()
    gi, gd = idx[:N].cpu().numpy(), dist[:N].cpu().numpy()
    fin = np.isfinite(gd)
    rep = parity.compare(gi[fin], gd[fin], o[0][fin], o[1][fin], o[2][fin], atol=2e-6)
    assert rep.ok, str(rep)
    assert not fin[5 % N] and gi[5 % N] == 0 and not fin[40 % N] and gi[40 % N] == 0
    assert fin.sum() >= N - 2
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    ctx.set_labels(cls)
    jp, dp, jn, dn = ctx.assign_matched(X[:N], y)
    assert ctx.last_kernel == "mfma_omega_two_pass"
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        om = oracle.assign(X[:N], W, omega=Om, proto_cls=cls, labels=y, matching=matching, threads=8)
        f2 = np.isfinite(d) | ~np.isfinite(om[1])
        f2 &= fin | ~np.isfinite(om[1])
        rep = parity.compare(j[f2], d[f2], om[0][f2], om[1][f2], om[2][f2], atol=2e-6)
        assert rep.ok, f"{case} matching={matching}: {rep}"
    Xc = X[:N].copy()
    Xc[~np.isfinite(Xc)] = 0.5
    Z = ctx.project(Xc, PROJECT_OMEGA)
    assert ctx.last_kernel == ("mfma_project_long" if D > 256 else "mfma_project")
    assert _rel_err(Z, oracle.project_omega(Om, Xc)) < 1e-5
    Y = ctx.project(Xc, PROJECT_LAMBDA)           # long rows and more than 128 Omega rows: Omega x, then Omega^T y
    assert ctx.last_kernel == ("mfma_project_two_pass" if D > 256 else "mfma_project")
    assert _rel_err(Y, oracle.project_lambda(Om, Xc)) < 1e-5


@pytest.mark.parametrize("case", [(600, 784, 10, 10), (300, 300, 3, 40), (200, 512, 5, 33), (65, 1000, 1, 3), (500, 257, 37, 2), (150, 400, 2, 100)],
                         ids=lambda c: "n%d_d%d_k%d_m%d" % c)
def test_long_rows_local(ctx, case):
    """LGMLVQ over rows above 256 values (k_local_long: the tiles of all Omega_j stacked, groups of up to 4 tiles, the
    rows walked once per group): plain and matched against the oracle, NaN / INF rows, nothing written past N."""
    import torch
    N, D, K, M = case
    rng = np.random.default_rng(N + D + K + M)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Oms = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    X[5 % N, D - 1] = np.nan
    X[40 % N, 0] = np.inf
    o = oracle.assign(X[:N], W, omega=Oms, threads=8)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    ctx.set_local_omegas(Oms)
    Xd = torch.from_numpy(X).cuda()
    idx = torch.full((N + 2,), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((N + 2,), -7.0, dtype=torch.float32, device="cuda")
    ctx.assign_into(Xd[:N], idx[:N], dist[:N])
    torch.cuda.synchronize()
    assert ctx.last_kernel == "mfma_local_long"
    assert (idx[N:].cpu().numpy() == -7).all() and (dist[N:].cpu().numpy() == -7.0).all()
    gi, gd = idx[:N].cpu().numpy(), dist[:N].cpu().numpy()
    fin = np.isfinite(gd)
    rep = parity.compare(gi[fin], gd[fin], o[0][fin], o[1][fin], o[2][fin], atol=2e-6)
    assert rep.ok, str(rep)
    assert not fin[5 % N] and gi[5 % N] == 0 and not fin[40 % N] and gi[40 % N] == 0
    assert fin.sum() >= N - 2
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    ctx.set_labels(cls)
    jp, dp, jn, dn = ctx.assign_matched(X[:N], y)
    assert ctx.last_kernel == "mfma_local_long"
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        om = oracle.assign(X[:N], W, omega=Oms, proto_cls=cls, labels=y, matching=matching, threads=8)
        f2 = np.isfinite(d) | ~np.isfinite(om[1])
        f2 &= fin | ~np.isfinite(om[1])
        rep = parity.compare(j[f2], d[f2], om[0][f2], om[1][f2], om[2][f2], atol=2e-6)
        assert rep.ok, f"{case} matching={matching}: {rep}"
    ctx.set_kernel_policy(KERNEL_SCALAR)                                  # the scalar kernel agrees
    si, sd = ctx.assign(X[:N])
    assert ctx.last_kernel == "scalar"
    ctx.set_kernel_policy(KERNEL_AUTO)
    same = fin & np.isfinite(sd)
    assert (si[same] == gi[same]).mean() > 0.999


@pytest.mark.parametrize("case", [(600, 784, 40, 16), (300, 300, 100, 40), (200, 512, 10, 128), (40, 1000, 257, 3)],
                         ids=lambda c: "n%d_d%d_k%d_m%d" % c)
def test_long_rows_omega(ctx, case):
    """Global Omega over rows above 256 values (k_omega_long: projection in chunks of 64, search in the projected
    space as in k_omega_pipe): plain and matched against the oracle, NaN / INF rows, nothing written past N."""
    import torch
    N, D, K, M = case
    rng = np.random.default_rng(N + D + K + M)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N + 2, D)).astype(np.float32)
    X[5 % N, D - 1] = np.nan
    X[17 % N, 300 % D] = np.inf
    o = oracle.assign(X[:N], W, omega=Om, threads=8)
    ctx.set_kernel_policy(KERNEL_AUTO)
    ctx.set_prototypes(W)
    ctx.set_omega(Om)
    Xd = torch.from_numpy(X).cuda()
    idx = torch.full((N + 2,), -7, dtype=torch.int32, device="cuda")
    dist = torch.full((N + 2,), -7.0, dtype=torch.float32, device="cuda")
    ctx.assign_into(Xd[:N], idx[:N], dist[:N])
    torch.cuda.synchronize()
    assert ctx.last_kernel == "mfma_omega_long"
    assert (idx[N:].cpu().numpy() == -7).all() and (dist[N:].cpu().numpy() == -7.0).all()
    gi, gd = idx[:N].cpu().numpy(), dist[:N].cpu().numpy()
    rep = parity.compare(gi, gd, *o, atol=2e-6)
    print(case, rep)
    assert rep.ok, str(rep)
    assert gi[5 % N] == 0 and np.isinf(gd[5 % N]) and gi[17 % N] == 0 and np.isinf(gd[17 % N])
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    ctx.set_labels(cls)
    jp, dp, jn, dn = ctx.assign_matched(X[:N], y)
    assert ctx.last_kernel == "mfma_omega_long"
    for matching, j, d in ((True, jp, dp), (False, jn, dn)):
        rep = parity.compare(j, d, *oracle.assign(X[:N], W, omega=Om, proto_cls=cls, labels=y, matching=matching, threads=8), atol=2e-6)
        assert rep.ok, f"{case} matching={matching}: {rep}"
