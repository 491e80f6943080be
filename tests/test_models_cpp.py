"""The C++ host mirror of the reference's model types (include/vq_models.hpp) and its test program
(tests/cpp/test_models.cpp, shaped like the reference's tests/*.rs).  The program is built in-tree by
`make` (plain g++ against the C ABI) and travels to the GPU box with the snapshot."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_models")


def test_models_program_is_built_and_lists_its_tests():
    assert os.path.exists(BIN), "run `make` (or __graft_entry__.build()) first"
    out = subprocess.run([BIN, "--list"], check=True, capture_output=True, text=True, timeout=60).stdout.split()
    for model in ("vq", "lvq", "glvq", "gmlvq", "lgmlvq", "liramlvq"):
        assert f"check_constraints_predict_{model}" in out
    assert "renaming_prototypes" in out and "simple_clustering_vq" in out


def test_header_is_self_contained_cxx17(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "vq_models.hpp"\nint main() { vq::GLVQ m; (void)m; return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", f"-I{ROOT}/include", str(src)],
                   check=True, timeout=120)


@pytest.mark.gpu
def test_models_program_passes_on_the_gpu():
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]


@pytest.mark.gpu
def test_models_program_linked_against_the_static_archive_passes_on_the_gpu():
    """north_star: "a hipcc-built static library linked from build.rs".  The same test program, linked by plain g++
    against vq_amd/libvqhip.a + libamdhip64 (Makefile: tests/cpp/test_models_static), as a Rust build would link it."""
    exe = BIN + "_static"
    assert os.path.exists(exe), "run `make` first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]
