"""ctypes handles on the two TEST-INFRASTRUCTURE libraries (never imported by libcimbar_amd):

  oracle/libcimbar_oracle.so       plain-C restatement (oracle/cimbar_oracle.c)        -> `oracle_lib()`
  oracle/_ref/libcimbar_ref.so     the reference's own sources + cv-shim (oracle/Makefile) -> `ref_lib()` (None if absent)
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libcimbar_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libcimbar_ref.so")


class CoCcm(ctypes.Structure):
    _fields_ = [("m", ctypes.c_float * 9), ("active", ctypes.c_int)]


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_oracle = None
_ref = None


def build_oracle():
    srcs = [os.path.join(HERE, "cimbar_oracle.c"), os.path.join(HERE, "cimbar_oracle_extract.c")]
    hdr = os.path.join(HERE, "cimbar_oracle.h")
    if os.path.exists(ORACLE_SO) and os.path.getmtime(ORACLE_SO) >= max(os.path.getmtime(p) for p in srcs + [hdr]):
        return ORACLE_SO
    subprocess.run(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-std=gnu11", "-Wall", "-shared", "-o", ORACLE_SO, *srcs, "-lm"], check=True)
    return ORACLE_SO


def oracle_lib():
    global _oracle
    if _oracle is None:
        build_oracle()
        L = ctypes.CDLL(ORACLE_SO)
        L.co_last_symbols.restype = ctypes.POINTER(ctypes.c_uint8)
        L.co_last_colors.restype = ctypes.POINTER(ctypes.c_uint8)
        L.co_last_positions.restype = ctypes.POINTER(ctypes.c_int32)
        L.co_best_color.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        L.co_best_color.restype = ctypes.c_uint
        _oracle = L
    return _oracle


def ref_lib():
    """The reference build, or None when oracle/_ref has not been built (it needs /root/reference at build time)."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        L = ctypes.CDLL(REF_SO)
        L.ref_sink_decode_frame.restype = ctypes.c_int64
        L.ref_best_color.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.ref_configure(68)
        _ref = L
    return _ref


# ---------------------------------------------------------------------------------------------- convenience wrappers
def oracle_decode(rgb, preprocess=0, cc=2, ccm=None):
    """co_decode_fountain on one (1024,1024,3) uint8 frame -> (good_bytes, chunks (12,625), mask, ccm struct)."""
    L = oracle_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    chunks = np.zeros((12, 625), np.uint8)
    mask = ctypes.c_uint32(0)
    if ccm is None:
        ccm = CoCcm()
    r = L.co_decode_fountain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), ctypes.byref(ccm), P(chunks), ctypes.byref(mask))
    return r, chunks, mask.value, ccm


def oracle_decode_plain(rgb, preprocess=0, cc=2, ccm=None):
    """co_decode_plain (Decoder::decode, the --no-fountain path) -> (ret, bytes (7500,), block_ok (60,), ccm struct)."""
    L = oracle_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    data = np.zeros(7500, np.uint8)
    ok = np.zeros(60, np.uint8)
    if ccm is None:
        ccm = CoCcm()
    r = L.co_decode_plain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), ctypes.byref(ccm), P(data), P(ok))
    return r, data, ok, ccm


def ref_decode_plain(rgb, preprocess=0, cc=2, reset_ccm=1):
    L = ref_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    data = np.zeros(7500, np.uint8)
    r = L.ref_decode_plain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), int(reset_ccm), P(data))
    return r, data


def oracle_stage(rgb_unused=None):
    """symbols, colours, drifted positions of the last oracle_decode call on this thread."""
    L = oracle_lib()
    sym = np.ctypeslib.as_array(L.co_last_symbols(), shape=(12400,)).copy()
    col = np.ctypeslib.as_array(L.co_last_colors(), shape=(12400,)).copy()
    pos = np.ctypeslib.as_array(L.co_last_positions(), shape=(12400, 2)).copy()
    return sym, col, pos


def ref_decode(rgb, preprocess=0, cc=2, reset_ccm=1):
    L = ref_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    chunks = np.zeros((12, 625), np.uint8)
    mask = ctypes.c_uint32(0)
    r = L.ref_decode_fountain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), int(reset_ccm), P(chunks), ctypes.byref(mask))
    return r, chunks, mask.value


def ref_encode_raw(payload):
    L = ref_lib()
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    rgb = np.zeros((1024, 1024, 3), np.uint8)
    L.ref_encode_raw(P(payload), payload.size, P(rgb))
    return rgb
