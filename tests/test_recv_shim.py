"""CPU: libcimbar_recv_hip.so loads and exports every symbol include/cimbar_recv_hip.h declares; the calls that need no device answer like the
reference's (cimbard_get_bufsize per mode, the -1 / -2 argument checks of cimbar_recv_js.cpp:150-159), and without a GPU the decode call fails
loudly with -4 (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "libcimbar_amd", "libcimbar_recv_hip.so")


def load():
    if not os.path.exists(LIB):
        pytest.fail("libcimbar_recv_hip.so not built: run `python -m libcimbar_amd.build` (or __graft_entry__.build())")
    return ctypes.CDLL(LIB)


def test_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "cimbar_recv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(cimbard_[a-z_]+)\s*\(", text)))
    assert names == ["cimbard_configure_decode", "cimbard_get_bufsize", "cimbard_get_report", "cimbard_hip_set_device", "cimbard_scan_extract_decode"]
    lib = load()
    for n in names:
        assert hasattr(lib, n), n


def test_bufsize_follows_the_configured_mode_like_the_reference(ref):
    lib = load()
    for mode in (68, 67, 66, 4, 8, 0, -3, 5, 68):
        assert lib.cimbard_configure_decode(mode) == 0 and ref.cimbard_configure_decode(mode) == 0
        assert lib.cimbard_get_bufsize() == ref.cimbard_get_bufsize()
    ref.ref_configure(68)


def test_argument_checks_and_no_cpu_fallback():
    lib = load()
    lib.cimbard_configure_decode(68)
    img = np.zeros(64 * 48 * 3, np.uint8)
    buf = np.zeros(7500, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.cimbard_scan_extract_decode(p(img), 0, 48, 3, p(buf), buf.size) == -1
    assert lib.cimbard_scan_extract_decode(p(img), 64, 0, 3, p(buf), buf.size) == -1
    assert lib.cimbard_scan_extract_decode(p(img), 64, 48, 3, p(buf), 7499) == -2
    import torch
    if not torch.cuda.is_available():
        assert lib.cimbard_scan_extract_decode(p(img), 64, 48, 3, p(buf), buf.size) == -4
        rep = ctypes.create_string_buffer(256)
        n = lib.cimbard_get_report(rep, 256)
        assert b"cimbar_hip_create failed" in rep.raw[:n]
