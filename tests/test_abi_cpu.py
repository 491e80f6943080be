"""CPU: the C-ABI library loads, exports every symbol include/vqhip.h declares, and its
GPU-free entry points (synthetic host fill, error paths) behave.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from vq_amd import _lib, synth


def _declared():
    text = open(os.path.join(ROOT, "include", "vqhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(vqhip_\w+)\s*\(", text))


def test_header_and_binding_list_agree():
    assert _declared() == set(_lib.SIGNATURES)


def test_every_declared_symbol_is_exported():
    L = C.CDLL(_lib.LIB_PATH)
    for name in sorted(_declared()):
        assert getattr(L, name) is not None, name


def test_library_is_hip_for_gfx950():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob                      # the offload bundle targets gfx950 ...
    assert b"gfx942" not in blob and b"sm_" not in blob   # ... and nothing else
    assert _lib.lib().vqhip_version().decode().startswith("vqhip")


def test_null_context_is_an_error_not_a_crash():
    L = _lib.lib()
    assert L.vqhip_set_prototypes(None, None, 0, 0) == _lib.ERR_INVALID
    assert L.vqhip_assign(None, None, 0, None, None) == _lib.ERR_INVALID
    assert L.vqhip_destroy(None) == _lib.OK
    assert L.vqhip_last_kernel(None) == b"none"


def test_synth_host_is_counter_based_and_in_range():
    a = synth.values(4096, first=0)
    b = synth.values(1024, first=1000)
    assert (a[1000:2024] == b).all()              # any window can be regenerated
    assert a.min() >= -1.0 and a.max() < 1.0
    assert abs(a.mean()) < 0.05 and abs(a.std() - 1 / np.sqrt(3)) < 0.02
    assert not (synth.values(64, stream=1) == synth.values(64, stream=2)).any()
    assert not (synth.values(64, seed=1) == synth.values(64, seed=2)).all()
    # 24-bit grid
    assert (np.round((a + 1.0) * 8388608.0) == (a + 1.0) * 8388608.0).all()


def test_synth_model_shapes_and_sample_formula():
    m = synth.make_model(synth.CONFIGS["c1"])
    assert m.W.shape == (3, 4) and m.cls.tolist() == [0, 1, 2] and m.omega is None
    X = synth.samples_host(m, 0, 150)
    u = synth.values(150 * 4, stream=synth.STREAM_X).reshape(150, 4)
    want = np.float32(0.25) * u + m.W[np.arange(150) % 3]          # one rounding, like fmaf
    assert (X == want.astype(np.float32)).all()
    assert (synth.samples_host(m, 40, 10) == X[40:50]).all()
    assert synth.sample_labels(m, 3, 4).tolist() == [0, 1, 2, 0]

    small = synth.Config("t", "GMLVQ", 10, 8, 4, M=8, omega="global")
    g = synth.make_model(small)
    assert g.omega.shape == (8, 8) and abs((g.omega.astype(np.float64) ** 2).sum() - 1.0) < 1e-6
    loc = synth.make_model(synth.Config("t", "LGMLVQ", 10, 8, 3, M=8, omega="local"))
    assert loc.omega.shape == (3, 8, 8)
    lira = synth.make_model(synth.Config("t", "LiRaMLVQ", 10, 8, 3, M=2, omega="global"))
    assert lira.omega.shape == (2, 8)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_create_without_gpu_fails_loudly():
    from vq_amd import Context, VqhipError

    with pytest.raises(VqhipError) as e:
        Context(0)
    assert e.value.code in (_lib.ERR_HIP, _lib.ERR_UNSUPPORTED)
    assert "HIP device" in e.value.text or "gfx950" in e.value.text


def test_host_side_builders_under_asan_and_ubsan():
    """the host half of the model builder (duplicate detection, centring, every tile-image builder)
    under -fsanitize=address,undefined, CPU only (`make asan` builds tests/cpp/test_host_asan with hipcc)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_host_asan")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", root, "asan"], check=True, timeout=900)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert r.returncode == 0 and "host_asan_ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
