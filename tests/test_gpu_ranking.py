"""GPU: ranking the candidates on the bf16 matrix cores changes nothing that leaves the library.

The LDS-resident kernels rank prototypes by ||w||^2 - 2 x.w with every f32 operand split into two bf16 pieces and three
piece products on v_mfma_f32_32x32x16_bf16 (mfma_device.h search_tiles_bf); vqhip_set_option(VQHIP_OPT_RANKING,
VQHIP_VALUE_F32) keeps the ranking on v_mfma_f32_32x32x2_f32.  Either way the best two candidates are re-measured in f32 in
the reference's direct form (src/helpers.rs:18-23) and a third key within the ranking's margin sends the sample to the
exact in-order scan (src/helpers.rs:55-77), so winners AND distances must agree bit for bit between the two settings --
on easy data, on dense clusters where the guard does the work, on wide-range data where the bf16 pieces are at their
worst, plain and matched.  (Against the f64 oracle both are held to the usual bar by the other GPU tests, which run with
the default, bf16 ranking.)"""
import zlib

import numpy as np
import pytest

import oracle
import parity
from vq_amd import KERNEL_AUTO, OPT_PROJECTION, OPT_RANKING, VALUE_BF16, VALUE_F32, Context

# option settings of a run (vqhip_set_option): everything on the f32 cores / ranking on bf16, projection f32 / both on bf16
ALL_F32 = {OPT_RANKING: VALUE_F32, OPT_PROJECTION: VALUE_F32}
RANK_BF16 = {OPT_RANKING: VALUE_BF16, OPT_PROJECTION: VALUE_F32}
ALL_BF16 = {OPT_RANKING: VALUE_BF16, OPT_PROJECTION: VALUE_BF16}

pytestmark = pytest.mark.gpu


def _run(X, W, Om, cls, y, options):
    ctx = Context(0)
    try:
        ctx.set_kernel_policy(KERNEL_AUTO)
        for opt, value in options.items():
            ctx.set_option(opt, value)
        ctx.set_prototypes(W)
        if Om is not None:
            ctx.set_omega(Om)
        idx, dist = ctx.assign(X)
        out = {"kernel": ctx.last_kernel, "ranking": ctx.last_ranking, "projection": ctx.last_projection, "idx": idx, "dist": dist}
        ctx.set_labels(cls)
        out["matched"] = ctx.assign_matched(X, y)
        assert ctx.last_ranking == out["ranking"] and ctx.last_projection == out["projection"]
        return out
    finally:
        ctx.close()


def _data(kind, rng, N, K, D):
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(K, D)).astype(np.float32), rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    if kind == "wide":          # twelve binary orders of magnitude inside every row, signs mixed
        W = (np.exp(rng.uniform(-4, 4, size=(K, D))) * rng.choice([-1.0, 1.0], size=(K, D))).astype(np.float32)
        X = (np.exp(rng.uniform(-4, 4, size=(N, D))) * rng.choice([-1.0, 1.0], size=(N, D))).astype(np.float32)
        return W, X
    if kind == "dense":         # clusters of 4 prototypes 1e-3 ||w|| apart, samples 3e-3 ||w|| from a cluster
        centres = rng.uniform(-1, 1, size=((K + 3) // 4, D))
        W = np.repeat(centres, 4, axis=0)[:K]
        u = rng.normal(size=(K, D))
        W = (W + 1e-3 * np.linalg.norm(W, axis=1, keepdims=True) * u / np.linalg.norm(u, axis=1, keepdims=True)).astype(np.float32)
        c = centres[rng.integers(0, len(centres), size=N)]
        v = rng.normal(size=(N, D))
        X = (c + 3e-3 * np.linalg.norm(c, axis=1, keepdims=True) * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        return W, X
    if kind == "grid":          # integer features: exact ties between distinct prototypes, the lowest index wins
        return rng.integers(0, 3, size=(K, D)).astype(np.float32), rng.integers(0, 3, size=(N, D)).astype(np.float32)
    raise ValueError(kind)


SHAPES = [
    # name, D, K, M (0: Euclidean), kernel
    ("glvq_c2", 128, 256, 0, "mfma_euclid_lds_pipe"),
    ("lvq_k20_staged", 128, 20, 0, "mfma_euclid_lds_staged"),
    ("glvq_d100_cut", 100, 37, 0, "mfma_euclid_lds_pipe_cut"),
    ("glvq_d16", 16, 300, 0, "mfma_euclid_lds_pipe"),
    ("vq_d32_k800_pieces_image", 32, 800, 0, "mfma_euclid_lds_pipe"),       # the LDS image holds bf16 pieces (BFIMG), R = 32
    ("vq_d64_k400_pieces_image", 64, 400, 0, "mfma_euclid_lds_pipe"),       # the same at R = 64
    ("glvq_d120_k200_cut_pieces_image", 120, 200, 0, "mfma_euclid_lds_pipe_cut"),
    ("gmlvq_c3", 128, 256, 128, "mfma_omega_lds"),
    ("liramlvq_c4", 256, 1024, 16, "mfma_omega_lowrank_lds"),
    ("vq_k1024_streamed", 128, 1024, 0, "mfma_euclid"),
    ("gmlvq_k600_streamed", 128, 600, 128, "mfma_omega"),
]


@pytest.mark.parametrize("kind", ["uniform", "wide", "dense", "grid"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_bf16_ranking_returns_the_f32_rankings_results(shape, kind):
    name, D, K, M, kernel = shape
    rng = np.random.default_rng(zlib.crc32(f"{name}-{kind}".encode()))
    N = 4099
    W, X = _data(kind, rng, N, K, D)
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32) if M else None
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    # the projection is part of the result (vqhip_last_projection), the ranking is not: the rankings are compared with
    # the projection held on the f32 matrix cores; the bf16x6 projection is checked against the oracle below
    bf = _run(X, W, Om, cls, y, RANK_BF16)
    f32 = _run(X, W, Om, cls, y, ALL_F32)
    assert bf["kernel"] == kernel and f32["kernel"] == kernel, (bf["kernel"], f32["kernel"])
    assert bf["ranking"] == "bf16x3" and f32["ranking"] == "f32", (bf["ranking"], f32["ranking"])
    assert bf["projection"] == f32["projection"] == ("f32" if M else "none")
    assert np.array_equal(bf["idx"], f32["idx"]), f"{int((bf['idx'] != f32['idx']).sum())} winners differ"
    assert np.array_equal(bf["dist"].view(np.uint32), f32["dist"].view(np.uint32)), "distances differ"
    for a, b, what in zip(bf["matched"], f32["matched"], ("j+", "d+", "j-", "d-")):
        assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)), f"matched {what} differs"
    if kernel == "mfma_omega_lds":      # the default: ranking AND projection on the bf16 matrix cores
        bf = _run(X, W, Om, cls, y, ALL_BF16)
        assert bf["kernel"] == kernel and bf["ranking"] == "bf16x3" and bf["projection"] == "bf16x6", bf
        # against the f32 projection: the winner's distance agrees to the oracle's bar whichever index carries it, and
        # the index differs only where two prototypes are that close (near-ties)
        def same(j, d, fj, fd, what):
            d, fd = np.asarray(d, dtype=np.float64), np.asarray(fd, dtype=np.float64)
            fin = np.isfinite(fd)
            assert np.array_equal(np.isfinite(d), fin), what
            assert np.all(np.abs(d[fin] - fd[fin]) <= 2e-5 * np.abs(fd[fin]) + 2e-6), (what, float(np.abs(d[fin] - fd[fin]).max()))
            assert int((np.asarray(j) != np.asarray(fj)).sum()) <= max(2, N // 500), what
        same(bf["idx"], bf["dist"], f32["idx"], f32["dist"], "plain")
        same(bf["matched"][0], bf["matched"][1], f32["matched"][0], f32["matched"][1], "matched +")
        same(bf["matched"][2], bf["matched"][3], f32["matched"][2], f32["matched"][3], "matched -")
    # and the shared answer is the oracle's (a subset keeps the f64 oracle quick on the Omega shapes)
    sel = slice(0, 400 if M else N)
    rep = parity.compare(bf["idx"][sel], bf["dist"][sel], *oracle.assign(X[sel], W, omega=Om, threads=8), atol=2e-6 if M else 0.0)
    print(name, kind, rep)
    assert rep.ok, str(rep)
    if kind == "grid" and not M:
        assert rep.mismatches == 0, str(rep)


LOCAL_SHAPES = [(128, 8, 128), (64, 8, 64), (256, 5, 40), (200, 3, 33), (32, 16, 32), (10, 5, 10)]


@pytest.mark.parametrize("shape", LOCAL_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_local_projection_on_the_bf16_cores_against_the_f32_engine(shape):
    """LGMLVQ (src/lgmlvq/lgmlvq.rs:436-506): k_local_pj (six bf16 piece products per product) against k_engine<LOCAL> (f32 MFMA,
    VQHIP_OPT_PROJECTION = VQHIP_VALUE_F32) on the same model -- distances to rounding, winners up to near-ties -- and both against the f64 oracle."""
    D, K, M = shape
    rng = np.random.default_rng(zlib.crc32(f"local-{D}-{K}-{M}".encode()))
    N = 2500
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Oms = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    cls = (np.arange(K) % 3).astype(np.uint32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)

    def run(options):
        ctx = Context(0)
        try:
            ctx.set_kernel_policy(KERNEL_AUTO)
            ctx.set_prototypes(W)
            ctx.set_local_omegas(Oms)
            for opt, value in options.items():            # after the model: setting an option drops the images like any set_*
                ctx.set_option(opt, value)
            idx, dist = ctx.assign(X)
            out = (ctx.last_kernel, ctx.last_projection, idx, dist)
            ctx.set_labels(cls)
            out += (ctx.assign_matched(X, y),)
            return out
        finally:
            ctx.close()

    kb, pb, ib, db, mb = run(ALL_BF16)
    kf, pf, i32, d32, m32 = run(RANK_BF16)
    assert (kb, pb) == ("mfma_local_bf16", "bf16x6") and (kf, pf) == ("mfma_local", "f32"), (kb, pb, kf, pf)
    for j, d, fj, fd in ((ib, db, i32, d32), (mb[0], mb[1], m32[0], m32[1]), (mb[2], mb[3], m32[2], m32[3])):
        d, fd = np.asarray(d, dtype=np.float64), np.asarray(fd, dtype=np.float64)
        fin = np.isfinite(fd)
        assert np.array_equal(np.isfinite(d), fin)
        assert np.all(np.abs(d[fin] - fd[fin]) <= 2e-5 * np.abs(fd[fin]) + 2e-6), float(np.abs(d[fin] - fd[fin]).max())
        assert int((np.asarray(j) != np.asarray(fj)).sum()) <= 2
    sel = slice(0, 500)
    for idx, dist in ((ib, db), (i32, d32)):
        rep = parity.compare(idx[sel], dist[sel], *oracle.assign(X[sel], W, omega=Oms, threads=8), atol=2e-6)
        print(shape, rep)
        assert rep.ok and rep.floor_used == 0, str(rep)
