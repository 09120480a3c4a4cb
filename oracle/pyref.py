"""ctypes handles on the two TEST-INFRASTRUCTURE libraries (never imported by libcimbar_amd):

  oracle/libcimbar_oracle.so       plain-C restatement (oracle/cimbar_oracle.c)        -> `oracle_lib()` (mode B; `oracle_lib(67)` = libcimbar_oracle_m67.so)
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


_oracle = {}
_ref = None

# mode -> (image w, image h, cells, chunk bytes, RS blocks per frame, RS data bytes per block, chunks per frame): Config.h:101-165 for the modes
# the oracle is built for
GEOMETRY = {68: (1024, 1024, 12400, 625, 60, 125, 12), 67: (1024, 720, 8592, 429, 36, 143, 12), 66: (736, 637, 5376, 540, 24, 135, 6),
            4: (1024, 1024, 12400, 750, 60, 125, 10), 8: (1024, 1024, 12400, 875, 70, 125, 10)}


def oracle_so(mode=68):
    return ORACLE_SO if mode == 68 else os.path.join(HERE, "libcimbar_oracle_m%d.so" % mode)


def build_oracle(mode=68):
    srcs = [os.path.join(HERE, "cimbar_oracle.c"), os.path.join(HERE, "cimbar_oracle_extract.c")]
    hdr = os.path.join(HERE, "cimbar_oracle.h")
    so = oracle_so(mode)
    if os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(p) for p in srcs + [hdr]):
        return so
    subprocess.run(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-std=gnu11", "-Wall", "-DCO_MODE=%d" % mode, "-shared", "-o", so, *srcs, "-lm"], check=True)
    return so


def oracle_lib(mode=68):
    """The C restatement built for one mode (68 = "B", 67 = "Bm", 66 = "Bu"): one geometry per library, see cimbar_oracle.h."""
    if mode not in _oracle:
        L = ctypes.CDLL(build_oracle(mode))
        L.co_last_symbols.restype = ctypes.POINTER(ctypes.c_uint8)
        L.co_last_colors.restype = ctypes.POINTER(ctypes.c_uint8)
        L.co_last_positions.restype = ctypes.POINTER(ctypes.c_int32)
        L.co_best_color.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        L.co_best_color.restype = ctypes.c_uint
        _oracle[mode] = L
    return _oracle[mode]


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


class ref_mode:
    """`with ref_mode(67): ...` -- cimbar::Config::update(mode) on this thread for the block, back to mode B after (Config.h:46-50:
    the active conf is thread_local, so every ref_* call of the block must come from the same thread)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        ref_lib().ref_configure(self.mode)
        return ref_lib()

    def __exit__(self, *exc):
        ref_lib().ref_configure(68)
        return False


# ---------------------------------------------------------------------------------------------- convenience wrappers
def oracle_decode(rgb, preprocess=0, cc=2, ccm=None, mode=68):
    """co_decode_fountain on one (h,w,3) uint8 frame of `mode` -> (good_bytes, chunks (12,chunk), mask, ccm struct)."""
    L = oracle_lib(mode)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    chunks = np.zeros((GEOMETRY[mode][6], GEOMETRY[mode][3]), np.uint8)
    mask = ctypes.c_uint32(0)
    if ccm is None:
        ccm = CoCcm()
    r = L.co_decode_fountain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), ctypes.byref(ccm), P(chunks), ctypes.byref(mask))
    return r, chunks, mask.value, ccm


def oracle_decode_plain(rgb, preprocess=0, cc=2, ccm=None, mode=68):
    """co_decode_plain (Decoder::decode, the --no-fountain path) -> (ret, bytes (blocks*data,), block_ok (blocks,), ccm struct)."""
    L = oracle_lib(mode)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    g = GEOMETRY[mode]
    data = np.zeros(g[4] * g[5], np.uint8)
    ok = np.zeros(g[4], np.uint8)
    if ccm is None:
        ccm = CoCcm()
    r = L.co_decode_plain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), ctypes.byref(ccm), P(data), P(ok))
    return r, data, ok, ccm


def ref_decode_plain(rgb, preprocess=0, cc=2, reset_ccm=1):
    """mode B only (the wrapper's buffer is 7500 bytes)"""
    L = ref_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    data = np.zeros(7500, np.uint8)
    r = L.ref_decode_plain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), int(reset_ccm), P(data))
    return r, data


def oracle_stage(rgb_unused=None, mode=68):
    """symbols, colours, drifted positions of the last oracle_decode call on this thread."""
    L = oracle_lib(mode)
    n = GEOMETRY[mode][2]
    sym = np.ctypeslib.as_array(L.co_last_symbols(), shape=(n,)).copy()
    col = np.ctypeslib.as_array(L.co_last_colors(), shape=(n,)).copy()
    pos = np.ctypeslib.as_array(L.co_last_positions(), shape=(n, 2)).copy()
    return sym, col, pos


def ref_decode(rgb, preprocess=0, cc=2, reset_ccm=1, mode=68):
    """the caller has selected `mode` with ref_mode() already; it only sizes the chunk slots here"""
    L = ref_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    chunks = np.zeros((GEOMETRY[mode][6], GEOMETRY[mode][3]), np.uint8)
    mask = ctypes.c_uint32(0)
    r = L.ref_decode_fountain(P(rgb), rgb.shape[1], rgb.shape[0], int(preprocess), int(cc), int(reset_ccm), P(chunks), ctypes.byref(mask))
    return r, chunks, mask.value


def ref_encode_raw(payload, mode=68):
    L = ref_lib()
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    rgb = np.zeros((GEOMETRY[mode][1], GEOMETRY[mode][0], 3), np.uint8)
    L.ref_encode_raw(P(payload), payload.size, P(rgb))
    return rgb
