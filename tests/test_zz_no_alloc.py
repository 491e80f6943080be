"""GPU: vqhip_assign_device_noalloc (the entry point that allocates nothing) against vqhip_assign_device, through
the C ABI.  Same kernels, the batch counters from the library's file-scope __device__ array instead of the model's
hipMalloc'ed slots: outputs must be identical bit for bit, on two calls in a row (the second with a smaller input, the
counters still holding the first call's claims), for every kernel family that claims batches from a queue."""
import ctypes as C

import numpy as np
import pytest

import oracle
import parity
from vq_amd import KERNEL_AUTO, KERNEL_SCALAR, Context, VqhipError
from vq_amd._lib import ERR_INVALID, ERR_STATE, ERR_UNSUPPORTED

pytestmark = pytest.mark.gpu

NOALLOC_MAX_ROWS = 1 << 36

# name, D, K, M, kind, kernel family
MODELS = [
    ("glvq_c2_shape", 128, 256, 0, "none", "mfma_euclid_lds_pipe"),
    ("lvq_few_prototypes", 128, 20, 0, "none", "mfma_euclid_lds_staged"),
    ("glvq_ragged_rows", 100, 37, 0, "none", "mfma_euclid_lds_pipe_cut"),
    ("glvq_streamed", 128, 600, 0, "none", "mfma_euclid"),
    ("glvq_long_rows", 300, 50, 0, "none", "mfma_euclid_long"),
    ("gmlvq_c3_shape", 128, 256, 128, "global", "mfma_omega_lds"),
    ("liramlvq_c4_shape", 256, 1024, 16, "global", "mfma_omega_lowrank_lds"),
    ("lgmlvq_local", 128, 8, 128, "local", "mfma_local_bf16"),
]


def _load(ctx, D, K, M, kind, rng):
    W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
    Om = None
    ctx.set_prototypes(W)
    if kind == "global":
        Om = (rng.uniform(-1, 1, size=(M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
        ctx.set_omega(Om)
    elif kind == "local":
        Om = (rng.uniform(-1, 1, size=(K, M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
        ctx.set_local_omegas(Om)
    return W, Om


@pytest.mark.parametrize("model", MODELS, ids=lambda m: m[0])
def test_noalloc_equals_assign_device_twice(model):
    import torch

    name, D, K, M, kind, family = model
    rng = np.random.default_rng(len(name) + D + K)
    ctx = Context(0)
    try:
        ctx.set_kernel_policy(KERNEL_AUTO)
        W, Om = _load(ctx, D, K, M, kind, rng)
        ctx.prepare()
        # first call 20 001 rows (625 batches and a ragged one: every wave claims from the queue), second call 777
        for N in (20001, 777):
            X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
            Xd = torch.from_numpy(X).cuda()
            ref_i = torch.full((N + 5,), -7, dtype=torch.int32, device="cuda")
            ref_d = torch.full((N + 5,), -7.0, dtype=torch.float32, device="cuda")
            ctx.assign_into(Xd, ref_i[:N], ref_d[:N])                        # the existing entry point
            assert ctx.last_kernel == family
            got_i = torch.full((N + 5,), -7, dtype=torch.int32, device="cuda")
            got_d = torch.full((N + 5,), -7.0, dtype=torch.float32, device="cuda")
            ctx.assign_into_noalloc(Xd, got_i[:N], got_d[:N])
            assert ctx.last_kernel == family
            torch.cuda.synchronize()
            assert torch.equal(got_i, ref_i), f"{name} N={N}: indices differ"
            assert torch.equal(got_d.view(torch.int32), ref_d.view(torch.int32)), f"{name} N={N}: distances differ"
            assert (got_i[N:] == -7).all() and (got_d[N:] == -7.0).all()     # nothing written past N
            # and both are right: the parity bar against the oracle
            rows = np.unique(np.linspace(0, N - 1, 300 if kind != "none" and D * K > 100000 else 2000).astype(np.int64))
            rep = parity.compare(got_i[:N].cpu().numpy()[rows], got_d[:N].cpu().numpy()[rows],
                                 *oracle.assign(X[rows], W, omega=Om, threads=8), atol=0.0 if kind == "none" else 2e-6)
            assert rep.ok, f"{name} N={N}: {rep}"
    finally:
        ctx.close()


def test_noalloc_rejects_what_it_cannot_serve():
    import torch

    rng = np.random.default_rng(1)
    ctx = Context(0)
    try:
        X = torch.zeros((64, 128), dtype=torch.float32, device="cuda")
        idx = torch.zeros(64, dtype=torch.int32, device="cuda")
        with pytest.raises(VqhipError) as e:                               # nothing set
            ctx.D = 128
            ctx.assign_into_noalloc(X, idx)
        assert e.value.code == ERR_STATE
        _load(ctx, 128, 256, 0, "none", rng)
        with pytest.raises(VqhipError) as e:                               # set, not prepared: building would allocate
            ctx.assign_into_noalloc(X, idx)
        assert e.value.code == ERR_STATE and "prepare" in e.value.text
        ctx.prepare()
        ctx.assign_into_noalloc(X, idx)
        with pytest.raises(VqhipError) as e:                               # beyond the stated maximum (checked before anything is touched)
            ctx.assign_into_noalloc(X, idx, n=NOALLOC_MAX_ROWS + 1)
        assert e.value.code == ERR_INVALID and "VQHIP_NOALLOC_MAX_ROWS" in e.value.text
        ctx.assign_into_noalloc(X, idx, n=0)                               # empty input: nothing to do
        ctx.set_prototypes(rng.uniform(-1, 1, size=(5, 128)).astype(np.float32))
        with pytest.raises(VqhipError) as e:                               # a vqhip_set_* drops the images
            ctx.assign_into_noalloc(X, idx)
        assert e.value.code == ERR_STATE
        # a model that needs scratch proportional to N (two passes through projected rows)
        _load(ctx, 300, 6, 200, "global", rng)
        ctx.prepare()
        X3 = torch.zeros((64, 300), dtype=torch.float32, device="cuda")
        with pytest.raises(VqhipError) as e:
            ctx.assign_into_noalloc(X3, idx)
        assert e.value.code == ERR_UNSUPPORTED
        ctx.set_kernel_policy(KERNEL_SCALAR)
        with pytest.raises(VqhipError) as e:
            ctx.assign_into_noalloc(X3, idx)
        assert e.value.code == ERR_STATE
    finally:
        ctx.close()


def test_noalloc_calls_no_allocator():
    """The entry point must not grow the device heap: hipMemGetInfo before and after many calls on a prepared model."""
    import torch

    rng = np.random.default_rng(2)
    ctx = Context(0)
    try:
        _load(ctx, 128, 256, 0, "none", rng)
        ctx.prepare()
        X = torch.from_numpy(rng.uniform(-1, 1, size=(5000, 128)).astype(np.float32)).cuda()
        idx = torch.zeros(5000, dtype=torch.int32, device="cuda")
        dist = torch.zeros(5000, dtype=torch.float32, device="cuda")
        ctx.assign_into_noalloc(X, idx, dist)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for _ in range(50):
            ctx.assign_into_noalloc(X, idx, dist)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0
    finally:
        ctx.close()
