"""The C-ABI library loads and exports every symbol include/cimbar_hip.h declares (no compute: runs without a GPU)."""
import os
import re

import pytest

from libcimbar_amd import decoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "cimbar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cimbar_hip_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(decoder.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(decoder.LIB_PATH):
        pytest.fail("libcimbar_hip.so not built: run `python -m libcimbar_amd.build` (or __graft_entry__.build())")
    lib = decoder.load_library()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.cimbar_hip_bufsize() == 7500
    assert [lib.cimbar_hip_mode_bufsize(m) for m in (68, 67, 66, 4, 8, 0, 5)] == [7500, 5148, 3240, 7500, 8750, 7500, 7500]


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(decoder.CimbarHipError):
        decoder.HipDecoder(0)


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under libcimbar_amd/ or include/ may import, include, link or dlopen it
    pats = [r"^\s*(import|from)\s+oracle\b", r"#\s*include\s*[<\"][^>\"]*oracle", r"libcimbar_oracle", r"libcimbar_ref", r"\bpyref\b"]
    for base in ("libcimbar_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".inc", ".h", ".hpp", ".cpp")):
                    text = open(os.path.join(dirpath, fn), errors="replace").read()
                    for pat in pats:
                        assert not re.search(pat, text, flags=re.M), (fn, pat)


def test_cpp_adapter_and_its_test_program_compile():
    """the header-only C++ adapter (libcimbar_amd/host/Decoder.h) and the program the GPU test runs are at least well-formed C++17
    against include/cimbar_hip.h on any machine (the GPU box builds and runs them, tests/test_gpu_cpp_adapter.py)"""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")
    res = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", src], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    hdr = os.path.join(ROOT, "include", "cimbar_hip.h")
    res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr      # the public header is plain C


def test_tile_hashes_computed_by_the_library_match_the_reference_build(ref):
    """row a19: the library derives its 16 tile hashes at create time from embedded bitmaps (CimbDecoder.cpp:87-99); they must be the
    ones the reference's CimbDecoder constructor computes (ref_tile_hashes) and the golden table in libcimbar_amd/modeb.py"""
    import numpy as np
    from libcimbar_amd import modeb
    from oracle import pyref
    mine = decoder.tile_hashes()
    assert (mine == modeb.TILE_HASHES).all()
    want = np.zeros(16, np.uint64)
    assert ref.ref_tile_hashes(pyref.P(want)) == 16
    assert (mine == want).all()


def test_tile_hashes_match_the_golden_table():
    from libcimbar_amd import modeb
    assert (decoder.tile_hashes() == modeb.TILE_HASHES).all()
