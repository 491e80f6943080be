"""Rates on the inputs the in-kernel guard (third key, exact scan / crowded list) exists for -- not part of the contract.

    python tools/guard_bench.py [case ...]      # default: all cases; results as one line each

Cases ("the guard's price where it fires"):
  synth_c2      config 2's own clustered data (the uncrowded rate everything else is compared with)
  random_c2     uniform random prototypes and samples, K = 256, D = 128
  init_state    the reference's initial state (src/lvq/lvq.rs:120): 8 classes x 12 prototypes on a line +-0.1, D = 64
  clustered_256 / clustered_2048   clusters of 3-5 prototypes 1e-3 ||w|| apart, samples 3e-3 ||w|| from a cluster
  binary_256    0 / 1 features, K = 256, D = 128 (exact multi-way ties)
  random_1024 / random_4096        uniform random, streamed engine
  pixels_784 / pixels_1024         data in [0, 1), K = 50 / 100: the long-row kernels with the centre subtracted
  synth_c3 / random_c3, synth_c4 / random_c4   the Omega kernels (k_omega_pipe, k_lowrank_pipe: no finish pass, samples the
                screening cannot settle are scanned in their wave) on their own clustered data and on uniform random data
"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vq_amd import Context, synth


def clustered(rng, K, D, spread_rel):
    W = np.empty((K, D), dtype=np.float32)
    centres = []
    k = 0
    while k < K:
        m = min(int(rng.choice((3, 4, 5))), K - k)
        c = rng.uniform(-1, 1, size=D)
        step = spread_rel * np.linalg.norm(c)
        for _ in range(m):
            u = rng.normal(size=D)
            W[k] = (c + step * u / np.linalg.norm(u)).astype(np.float32)
            centres.append(c)
            k += 1
    return W, np.array(centres)


def make(case, rng, n_host):
    """-> (W, X_host block that is tiled on the device)"""
    if case in ("random_c3", "random_c4"):
        D, K, M = (128, 256, 128) if case == "random_c3" else (256, 1024, 16)
        Om = (rng.uniform(-1, 1, (M, D)) / np.sqrt(M * D / 3.0)).astype(np.float32)
        return rng.uniform(-1, 1, (K, D)).astype(np.float32), rng.uniform(-1, 1, (n_host, D)).astype(np.float32), Om
    if case in ("init_state_omega", "binary_c3", "clustered_c3"):
        # the Euclidean cases of the same names under a global Omega (k_omega_pipe): the GMLVQ initial state
        # (src/gmlvq/gmlvq.rs:160-161: Omega = I / sqrt(n)), binary features and dense clusters under a 128 x 128 Omega
        base = {"init_state_omega": "init_state", "binary_c3": "binary_256", "clustered_c3": "clustered_256"}[case]
        W, Xh = make(base, rng, n_host)[:2]
        D = W.shape[1]
        if case == "init_state_omega":
            Om = (np.eye(D) / np.sqrt(D)).astype(np.float32)
        else:
            Om = np.eye(D) + 0.1 * rng.uniform(-1, 1, (D, D))
            Om = (Om / np.linalg.norm(Om)).astype(np.float32)
        return W, Xh, Om
    if case in ("clustered_c4", "init_state_c4"):
        # LiRaMLVQ-shaped (k_lowrank_pipe): Omega 16 x D as src/liramlvq/liramlvq.rs:167-171 draws it
        if case == "clustered_c4":
            D, K = 256, 1024
            W, centres = clustered(rng, K, D, 1e-3)
            pick = rng.integers(0, K, size=n_host)
            u = rng.normal(size=(n_host, D))
            u /= np.linalg.norm(u, axis=1, keepdims=True)
            Xh = (centres[pick] + 3e-3 * np.linalg.norm(centres[pick], axis=1, keepdims=True) * u).astype(np.float32)
        else:
            W, Xh = make("init_state", rng, n_host)[:2]
            D = W.shape[1]
        Om = rng.uniform(-1, 1, (16, D))
        return W, Xh, (Om / np.linalg.norm(Om)).astype(np.float32)
    if case == "random_c2":
        return rng.uniform(-1, 1, (256, 128)).astype(np.float32), rng.uniform(-1, 1, (n_host, 128)).astype(np.float32)
    if case == "init_state":
        D, classes, per = 64, 8, 12
        means = rng.uniform(-1, 1, size=(classes, D))
        W = np.concatenate([means[c] + rng.uniform(-0.1, 0.1, size=(per, 1)) for c in range(classes)]).astype(np.float32)
        X = (means[rng.integers(0, classes, size=n_host)] + 0.3 * rng.uniform(-1, 1, size=(n_host, D))).astype(np.float32)
        return W, X
    if case == "init_state_784":
        # the reference's initial state on pixel-like rows: 10 classes x 3 prototypes on a line +-0.1 around class means in [0, 1)
        D, classes, per = 784, 10, 3
        means = rng.uniform(0, 1, size=(classes, D))
        W = np.concatenate([means[c] + rng.uniform(-0.1, 0.1, size=(per, 1)) for c in range(classes)]).astype(np.float32)
        X = np.clip(means[rng.integers(0, classes, size=n_host)] + 0.3 * rng.uniform(-1, 1, size=(n_host, D)), 0, 1).astype(np.float32)
        return W, X
    if case.startswith("clustered_"):
        K, D = int(case.split("_")[1]), 128
        W, centres = clustered(rng, K, D, 1e-3)
        pick = rng.integers(0, K, size=n_host)
        u = rng.normal(size=(n_host, D))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        return W, (centres[pick] + 3e-3 * np.linalg.norm(centres[pick], axis=1, keepdims=True) * u).astype(np.float32)
    if case == "binary_256":
        return (rng.uniform(size=(256, 128)) < 0.5).astype(np.float32), (rng.uniform(size=(n_host, 128)) < 0.5).astype(np.float32)
    if case.startswith("random_"):
        K = int(case.split("_")[1])
        return rng.uniform(-1, 1, (K, 128)).astype(np.float32), rng.uniform(-1, 1, (n_host, 128)).astype(np.float32)
    if case.startswith("pixels_"):
        D = int(case.split("_")[1])
        K = 50 if D == 784 else 100
        return rng.uniform(0, 1, (K, D)).astype(np.float32), rng.uniform(0, 1, (n_host, D)).astype(np.float32)
    raise SystemExit(f"unknown case {case}")


OMEGA_CASES = ["synth_c3", "random_c3", "init_state_omega", "binary_c3", "clustered_c3", "synth_c4", "random_c4", "init_state_c4", "clustered_c4"]
CASES = ["synth_c2", "random_c2", "init_state", "clustered_256", "clustered_2048", "binary_256", "random_1024", "random_4096",
         "pixels_784", "pixels_1024", "init_state_784"]


def main():
    cases = sys.argv[1:] or CASES
    ctx = Context(0)
    for case in cases:
        rng = np.random.default_rng(zlib.crc32(case.encode()))      # (str hashes differ from process to process)
        if case.startswith("synth_"):
            m = synth.make_model(synth.CONFIGS[case.split("_")[1]])
            synth.load_model(ctx, m)
            N = 4_000_000
            X = synth.samples_device(ctx, m, 0, N)
        else:
            n_host = 262_144
            made = make(case, rng, n_host)
            W, Xh = made[0], made[1]
            D = W.shape[1]
            reps = max(1, min(16, int(2.0e9 // (n_host * D * 4))))
            if W.shape[0] >= 2048:
                reps = min(reps, 4)
            ctx.set_prototypes(W)
            if len(made) > 2:
                ctx.set_omega(made[2])
            X = torch.from_numpy(Xh).cuda().repeat(reps, 1).contiguous()
            N = X.shape[0]
        idx = torch.empty(N, dtype=torch.int32, device="cuda")
        dist = torch.empty(N, dtype=torch.float32, device="cuda")
        ctx.assign_into(X, idx, dist)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        steps = int(os.environ.get("GUARD_STEPS", "3"))
        ev[0].record()
        for _ in range(steps):
            ctx.assign_into(X, idx, dist)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / steps
        print(f"{case:16s} kernel={ctx.last_kernel:26s} N={N:9d} K={ctx.K:5d} D={ctx.D:5d} ms={ms:9.3f} samples/s={N / ms * 1e3:.4g}", flush=True)
        del X, idx, dist
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
