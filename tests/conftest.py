import json
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _dec(o):
    if isinstance(o, str) and o in ("nan", "inf", "-inf"):
        return float(o)
    if isinstance(o, list):
        return [_dec(v) for v in o]
    if isinstance(o, dict):
        return {k: _dec(v) for k, v in o.items()}
    return o


def load_kat():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return _dec(json.load(f))


def intern_names(names):
    """Prototype / label names -> dense uint32 ids in sorted-name order (the BTreeMap order of
    the reference, src/lvq/lvq.rs:101).  Names absent from `names` can be mapped later with
    the returned dict's .get(name, 0xFFFFFFFF)."""
    table = {n: i for i, n in enumerate(sorted(set(names)))}
    return table


@pytest.fixture(scope="session")
def kat():
    return load_kat()


def assert_dist_close(got, want, rtol, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    for g, w in zip(got.ravel(), want.ravel()):
        if math.isinf(w) or math.isnan(w):
            assert (math.isinf(g) and g > 0) if math.isinf(w) else math.isnan(g), f"{what}: {g} vs {w}"
        else:
            assert abs(g - w) <= rtol * abs(w) + 1e-300 or (w == 0 and abs(g) <= 1e-30), f"{what}: {g} vs {w}"
