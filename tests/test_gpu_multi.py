"""GPU: the multi-device handle of the library (vqhip_multi_*), the pipelined host path of vqhip_project, and the matched
search over separately allocated f64 rows -- the host-pointer entry points a single-process caller (the Rust crate) uses.
A 1-GPU box lists device 0 twice: two contexts, two host threads, two shards."""
import numpy as np
import pytest

import oracle
import parity
from vq_amd import KERNEL_AUTO, PROJECT_LAMBDA, PROJECT_OMEGA, Context, MultiContext, VqhipError

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=lambda d: "x%d" % len(d))
def test_multi_handle_shards_and_matches_the_single_context(devices):
    rng = np.random.default_rng(len(devices))
    D, K, N = 128, 256, 50_001                                         # shards of unequal length
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    cls = (np.arange(K) % 4).astype(np.uint32)
    y = rng.integers(0, 4, size=N).astype(np.uint32)
    one = Context(0)
    m = MultiContext(devices)
    try:
        assert m.device_count == len(devices)
        one.set_prototypes(W); one.set_labels(cls)
        m.set_prototypes(W); m.set_labels(cls)
        i1, d1 = one.assign(X)
        i2, d2 = m.assign(X)
        assert (i1 == i2).all() and (d1.view(np.uint32) == d2.view(np.uint32)).all()
        rep = parity.compare(i2[:3000], d2[:3000], *oracle.assign(X[:3000], W, threads=8))
        assert rep.ok, str(rep)
        for a, b in zip(one.assign_matched(X, y), m.assign_matched(X, y)):
            assert (a.view(np.uint32) == b.view(np.uint32)).all()
        rows = [np.array(r, dtype=np.float64) for r in X[:5000]]
        i3, d3 = m.assign_rows(rows)
        assert (i3 == i1[:5000]).all() and (d3 == d1[:5000]).all()
        # fewer samples than devices, and none
        i4, _ = m.assign(X[:1])
        assert i4.tolist() == [int(i1[0])]
        assert m.assign(X[:0])[0].shape == (0,)
        # Omega models replicate too
        Om = (rng.uniform(-1, 1, size=(16, D)) / np.sqrt(16 * D / 3.0)).astype(np.float32)
        one.set_omega(Om); m.set_omega(Om)
        i5, d5 = one.assign(X[:9999]); i6, d6 = m.assign(X[:9999])
        assert (i5 == i6).all() and (d5 == d6).all()
        with pytest.raises(VqhipError) as e:                            # an error of one shard comes back with its device
            m.set_omega(np.zeros((3, D + 1), dtype=np.float32))
        assert e.value.code == -1 and "device 0" in e.value.text
    finally:
        m.close()
        one.close()
    with pytest.raises(VqhipError):
        MultiContext([99])


def test_multi_device_resident_shards_gather_over_rccl():
    """vqhip_multi_assign_device: device-resident shard -> search on the device's own stream -> RCCL exchange in place over the
    handle's communicator clique (ncclCommInitAll).  One GPU here: a clique of one rank (the gather is the identity), equal and
    "unequal" paths both run; the N > 1 exchange is unmeasured until an 8-GPU node runs it.  A device listed twice is refused
    (RCCL wants one rank per GPU)."""
    import torch

    rng = np.random.default_rng(21)
    D, K, N = 128, 256, 20_011
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    one, m = Context(0), MultiContext([0])
    try:
        one.set_prototypes(W)
        m.set_prototypes(W)
        i1, d1 = one.assign(X)
        Xd = torch.from_numpy(X).cuda()
        for _ in range(2):                                                # twice: streams and communicator are reused
            idx = torch.full((N,), -7, dtype=torch.int32, device="cuda")
            dist = torch.full((N,), -7.0, dtype=torch.float32, device="cuda")
            m.assign_device_shards([Xd], [idx], [dist])
            assert (idx.cpu().numpy().view(np.uint32) == i1).all()
            assert (dist.cpu().numpy().view(np.uint32) == d1.view(np.uint32)).all()
        idx = torch.full((N,), -7, dtype=torch.int32, device="cuda")
        m.assign_device_shards([Xd[:777]], [idx], None)                   # no distances; fewer rows than the arrays hold
        got = idx.cpu().numpy()
        assert (got[:777].view(np.uint32) == i1[:777]).all() and (got[777:] == -7).all()
        m.assign_device_shards([Xd[:0]], [idx], None)                     # nothing to do
    finally:
        m.close()
        one.close()
    m2 = MultiContext([0, 0])
    try:
        m2.set_prototypes(W)
        Xd = torch.from_numpy(X[:64]).cuda()
        idx = [torch.empty(128, dtype=torch.int32, device="cuda") for _ in range(2)]
        with pytest.raises(VqhipError) as e:
            m2.assign_device_shards([Xd, Xd], idx, None)
        assert e.value.code == -5 and "twice" in e.value.text
    finally:
        m2.close()


def test_project_host_path_is_pipelined_over_chunks():
    """vqhip_project from host pointers: more rows than one pinned buffer holds (D = 8: 4M-row chunks), both projections."""
    rng = np.random.default_rng(4)
    D, M = 8, 3
    N = (128 << 20) // (D * 4) * 2 + 777
    Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    ctx = Context(0)
    try:
        ctx.set_kernel_policy(KERNEL_AUTO)
        ctx.set_prototypes(rng.uniform(-1, 1, size=(3, D)).astype(np.float32))
        ctx.set_omega(Om)
        Z = ctx.project(X, PROJECT_OMEGA)
        Y = ctx.project(X, PROJECT_LAMBDA)
        for lo in (0, N // 2 - 300, N - 1000):
            want_z = oracle.project_omega(Om, X[lo:lo + 1000])
            want_y = oracle.project_lambda(Om, X[lo:lo + 1000])
            assert np.abs(Z[lo:lo + 1000] - want_z).max() <= 1e-5 * np.abs(want_z).max()
            assert np.abs(Y[lo:lo + 1000] - want_y).max() <= 1e-5 * np.abs(want_y).max()
    finally:
        ctx.close()


def test_matched_rows_f64_equals_contiguous_f32():
    rng = np.random.default_rng(6)
    D, K, N = 64, 70, 6000
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    cls = (np.arange(K) % 5).astype(np.uint32)
    y = rng.integers(0, 5, size=N).astype(np.uint32)
    ctx = Context(0)
    try:
        ctx.set_prototypes(W); ctx.set_labels(cls)
        a = ctx.assign_matched(X, y)
        b = ctx.assign_matched_rows([np.array(r, dtype=np.float64) for r in X], y)
        for u, v in zip(a, b):
            assert (u.view(np.uint32) == v.view(np.uint32)).all()
    finally:
        ctx.close()


def test_misaligned_device_pointers_are_refused():
    import torch

    ctx = Context(0)
    try:
        ctx.set_prototypes(np.zeros((4, 8), dtype=np.float32))
        raw = torch.zeros(8 * 16 + 1, dtype=torch.float32, device="cuda")
        idx = torch.zeros(16, dtype=torch.int32, device="cuda")
        import ctypes as C
        rc = ctx._L.vqhip_assign_device(ctx._h, C.c_void_p(raw.data_ptr() + 2), 16, C.c_void_p(idx.data_ptr()), None, None)
        assert rc == -1 and b"aligned" in ctx._L.vqhip_last_error(ctx._h)
    finally:
        ctx.close()


# ------------------------------------------------------------------ batched GLVQ gradient (second half)

def _glvq_reference(X, y, W, cls):
    """f64: the sums of src/glvq/glvq.rs:193-194 over all samples with the prototypes held fixed, from the oracle's matched
    searches (the reference itself updates after every sample: this is the batch form the ABI documents)."""
    jp, dp, _ = oracle.assign(X, W, proto_cls=cls, labels=y, matching=True, threads=8)
    jn, dn, _ = oracle.assign(X, W, proto_cls=cls, labels=y, matching=False, threads=8)
    dp, dn = np.asarray(dp, dtype=np.float64), np.asarray(dn, dtype=np.float64)
    s = dp + dn
    ok = np.isfinite(s) & (s > 0)
    X64, W64 = X.astype(np.float64), W.astype(np.float64)
    G = np.zeros_like(W64)
    cp = np.where(ok, -4.0 * dn / np.where(ok, s * s, 1.0), 0.0)
    cn = np.where(ok, 4.0 * dp / np.where(ok, s * s, 1.0), 0.0)
    np.add.at(G, jp[ok], cp[ok, None] * (X64[ok] - W64[jp[ok]]))
    np.add.at(G, jn[ok], cn[ok, None] * (X64[ok] - W64[jn[ok]]))
    cost = float(((dp[ok] - dn[ok]) / s[ok]).sum())
    return G, cost, int(ok.sum())


@pytest.mark.parametrize("shape", [(20_003, 128, 256), (5_000, 128, 600), (3_001, 4, 3), (4_000, 100, 37), (2_500, 300, 20)],
                         ids=lambda s: "n%d_d%d_k%d" % s)
def test_glvq_batch_gradient_against_f64_sums(shape):
    """vqhip_glvq_batch_gradient: matched search + scatter-add on the device (LDS partials up to ~36K prototype values, global
    atomics beyond: K = 600) against the same sums in f64 from the oracle's matched search.  f32 sums in no fixed order: 1e-4
    of the largest entry.  One sample in eight carries a label no prototype has: left out of gradient, cost and count."""
    import torch

    N, D, K = shape
    rng = np.random.default_rng(N + D + K)
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    cls = (np.arange(K) % 3).astype(np.uint32)
    X = (W[rng.integers(0, K, size=N)] + 0.4 * rng.uniform(-1, 1, size=(N, D))).astype(np.float32)
    y = rng.integers(0, 3, size=N).astype(np.uint32)
    y[::8] = 7                                                             # unknown class: no matching prototype
    want, want_cost, want_used = _glvq_reference(X, y, W, cls)
    tol = 1e-4 * np.abs(want).max()
    ctx = Context(0)
    try:
        ctx.set_kernel_policy(KERNEL_AUTO)
        ctx.set_prototypes(W)
        with pytest.raises(VqhipError) as e:
            ctx.glvq_batch_gradient(X, y)                                  # labels first
        assert e.value.code == -2
        ctx.set_labels(cls)
        for _ in range(2):                                                 # twice: the scratch is reused, the gradient overwritten
            G, cost, used = ctx.glvq_batch_gradient(X, y)
            assert used == want_used
            assert abs(cost - want_cost) <= 1e-4 * max(1.0, abs(want_cost))
            assert np.abs(G - want).max() <= tol, float(np.abs(G - want).max() / tol)
        Gd, tail = ctx.glvq_batch_gradient(torch.from_numpy(X).cuda(), torch.from_numpy(y.view(np.int32)).cuda())
        torch.cuda.synchronize()
        assert np.abs(Gd.cpu().numpy() - want).max() <= tol
        assert tail.cpu().numpy().view(np.uint64)[1] == want_used and abs(float(tail[0]) - want_cost) <= 1e-4 * max(1.0, abs(want_cost))
        G0, c0, u0 = ctx.glvq_batch_gradient(X[:0], y[:0])                  # no samples: a zero gradient
        assert (G0 == 0).all() and c0 == 0.0 and u0 == 0
        ctx.set_omega(np.eye(D, dtype=np.float32))
        with pytest.raises(VqhipError) as e:
            ctx.glvq_batch_gradient(X, y)                                  # Euclidean models only
        assert e.value.code == -5
    finally:
        ctx.close()
    m = MultiContext([0, 0])
    try:
        m.set_prototypes(W)
        m.set_labels(cls)
        G, cost, used = m.glvq_batch_gradient(X, y)                        # two shards, summed on the host
        assert used == want_used and np.abs(G - want).max() <= tol
    finally:
        m.close()


def test_matched_rows_f64_against_the_oracle_at_scale():
    """What vq::GLVQ::evaluate runs on (vqhip_assign_matched_rows_f64: the reference's container in, matched search on the device),
    12 000 samples against the oracle, with the GLVQ cost of src/glvq/glvq.rs:171-190 derived from it."""
    rng = np.random.default_rng(190)
    N, D, K = 12_000, 64, 96
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    cls = np.repeat(np.arange(8), 12).astype(np.uint32)                     # the reference's order: prototypes of a class together
    X = (W[rng.integers(0, K, size=N)] + 0.5 * rng.uniform(-1, 1, size=(N, D))).astype(np.float32)
    y = rng.integers(0, 8, size=N).astype(np.uint32)
    rows = [np.array(r, dtype=np.float64) for r in X]
    ctx = Context(0)
    try:
        ctx.set_prototypes(W)
        ctx.set_labels(cls)
        jp, dp, jn, dn = ctx.assign_matched_rows(rows, y)
        for matching, j, d in ((True, jp, dp), (False, jn, dn)):
            rep = parity.compare(j, d, *oracle.assign(X, W, proto_cls=cls, labels=y, matching=matching, threads=8))
            assert rep.ok, str(rep)
        _, want_cost, _ = _glvq_reference(X, y, W, cls)
        d1, d2 = dp.astype(np.float64), dn.astype(np.float64)
        assert abs(float(((d1 - d2) / (d1 + d2)).sum()) - want_cost) <= 1e-5 * abs(want_cost) + 1e-6
    finally:
        ctx.close()
