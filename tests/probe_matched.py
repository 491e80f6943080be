"""Probe (not collected by pytest; part of the test infrastructure, it uses the oracle): mismatches of the plain and
the labelled search against the oracle on uniform data of a given shape.  python tests/probe_matched.py D K N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from vq_amd import Context
import parity
import oracle
D, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(5)
W = rng.uniform(-1, 1, size=(K, D)).astype(np.float32)
X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
cls = (np.arange(K) % 3).astype(np.uint32)
y = rng.integers(0, 3, size=N).astype(np.uint32)
ctx = Context(0)
ctx.set_prototypes(W); ctx.set_labels(cls)
idx, dist = ctx.assign(X)
print("plain  ", ctx.last_kernel, parity.compare(idx, dist, *oracle.assign(X, W, threads=8)))
jp, dp, jn, dn = ctx.assign_matched(X, y)
for matching, j, d in ((True, jp, dp), (False, jn, dn)):
    o = oracle.assign(X, W, proto_cls=cls, labels=y, matching=matching, threads=8)
    print("matched", matching, ctx.last_kernel, parity.compare(j, d, *o))
