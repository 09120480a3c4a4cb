"""ctypes binding of include/cimbar_ingest.h (libcimbar_ingest.so): PNG decode pool + pinned ring + overlapped H2D in front of the device
decode path. Plumbing for tests / bench.py, like decoder.py."""
import ctypes
import os

import numpy as np

from . import decoder

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcimbar_ingest.so")
EXPORTS = ("cimbar_png_decode", "cimbar_ingest_create", "cimbar_ingest_destroy", "cimbar_ingest_last_error", "cimbar_ingest_run_files",
           "cimbar_ingest_run_raw", "cimbar_ingest_timings", "cimbar_ingest_create_ex", "cimbar_ingest_png_stats", "cimbar_jpeg_decode", "cimbar_ingest_fallback_overflow",
           "cimbar_image_decode", "cimbar_ingest_host_decoded")
PNG_HOST, PNG_DEVICE = 0, 1
SINK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int)

_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise decoder.CimbarHipError(f"{LIB_PATH} not found: build it with `python -m libcimbar_amd.build`")
        decoder.load_library()
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.cimbar_png_decode.argtypes = [vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
        L.cimbar_png_decode.restype = i32
        for fn in (L.cimbar_jpeg_decode, L.cimbar_image_decode):
            fn.argtypes = [vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
            fn.restype = i32
        L.cimbar_ingest_host_decoded.argtypes = [vp]
        L.cimbar_ingest_host_decoded.restype = ctypes.c_int64
        L.cimbar_ingest_create.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp)]
        L.cimbar_ingest_create.restype = i32
        L.cimbar_ingest_create_ex.argtypes = [vp, i32, i32, i32, i32, sz, ctypes.POINTER(vp)]
        L.cimbar_ingest_create_ex.restype = i32
        L.cimbar_ingest_png_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64)]
        L.cimbar_ingest_png_stats.restype = i32
        L.cimbar_ingest_destroy.argtypes = [vp]
        L.cimbar_ingest_destroy.restype = None
        L.cimbar_ingest_last_error.argtypes = [vp]
        L.cimbar_ingest_last_error.restype = ctypes.c_char_p
        L.cimbar_ingest_run_files.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), i32, i32, i32, SINK_FN, vp]
        L.cimbar_ingest_run_files.restype = ctypes.c_int64
        L.cimbar_ingest_run_raw.argtypes = [vp, vp, i32, i32, i32, SINK_FN, vp]
        L.cimbar_ingest_run_raw.restype = ctypes.c_int64
        L.cimbar_ingest_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.cimbar_ingest_timings.restype = i32
        _lib = L
    return _lib


def _decode_with(fn_name, data):
    L = load_library()
    fn = getattr(L, fn_name)
    buf = np.frombuffer(data, np.uint8)
    w, h = ctypes.c_uint(0), ctypes.c_uint(0)
    rc = fn(buf.ctypes.data, buf.size, None, 0, ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        raise decoder.CimbarHipError(f"{fn_name}: {rc}")
    out = np.zeros((h.value, w.value, 3), np.uint8)
    rc = fn(buf.ctypes.data, buf.size, out.ctypes.data, out.size, ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        raise decoder.CimbarHipError(f"{fn_name}: {rc}")
    return out


def jpeg_decode(data):
    """baseline JPEG bytes -> (h, w, 3) uint8 RGB with libjpeg's arithmetic (what cv::imread + BGR2RGB returns)"""
    return _decode_with("cimbar_jpeg_decode", data)


def image_decode(data):
    """PNG or JPEG, by the magic bytes"""
    return _decode_with("cimbar_image_decode", data)


def png_decode(data):
    """PNG bytes -> (h, w, 3) uint8 RGB, as cv::imread + BGR2RGB would hand it to the decoder"""
    L = load_library()
    buf = np.frombuffer(data, np.uint8)
    w, h = ctypes.c_uint(0), ctypes.c_uint(0)
    rc = L.cimbar_png_decode(buf.ctypes.data, buf.size, None, 0, ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        raise decoder.CimbarHipError(f"cimbar_png_decode: {rc}")
    out = np.zeros((h.value, w.value, 3), np.uint8)
    rc = L.cimbar_png_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.size, ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        raise decoder.CimbarHipError(f"cimbar_png_decode: {rc}")
    return out


class Ingest:
    def __init__(self, dec, threads=0, batch_frames=64, ring=3, png_device=False, zbytes_per_frame=0):
        """png_device: PNG files are inflated and un-filtered on the GPU (cimbar_ingest_create_ex, CIMBAR_INGEST_PNG_DEVICE)"""
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.cimbar_ingest_create_ex(dec._ctx, int(threads), int(batch_frames), int(ring), PNG_DEVICE if png_device else PNG_HOST,
                                               int(zbytes_per_frame), ctypes.byref(self._h))
        if rc != 0:
            raise decoder.CimbarHipError(f"cimbar_ingest_create_ex: {rc}")
        self._dec = dec

    def close(self):
        if self._h:
            self._lib.cimbar_ingest_destroy(self._h)
            self._h = ctypes.c_void_p()

    def _collect(self, n):
        chunks = np.zeros((n, self._dec.geo.FRAME_BYTES), np.uint8)
        masks = np.zeros(n, np.uint32)

        def cb(user, c, m, first, cnt):
            chunks[first:first + cnt] = np.ctypeslib.as_array(c, shape=(cnt, self._dec.geo.FRAME_BYTES))
            masks[first:first + cnt] = np.ctypeslib.as_array(m, shape=(cnt,))
            return 0
        return chunks, masks, SINK_FN(cb)

    def run_files(self, paths, should_preprocess=False, color_correction=2):
        n = len(paths)
        chunks, masks, cb = self._collect(n)
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        rc = self._lib.cimbar_ingest_run_files(self._h, arr, n, int(bool(should_preprocess)), int(color_correction), cb, None)
        if rc < 0:
            raise decoder.CimbarHipError(f"cimbar_ingest_run_files: {rc} {self._lib.cimbar_ingest_last_error(self._h).decode()}")
        return int(rc), chunks, masks

    def run_raw(self, frames, should_preprocess=False, color_correction=2, collect=True):
        frames = np.ascontiguousarray(frames, dtype=np.uint8) if not isinstance(frames, int) else frames
        n = frames.shape[0]
        if collect:
            chunks, masks, cb = self._collect(n)
        else:
            chunks = masks = None
            cb = SINK_FN(lambda user, c, m, first, cnt: 0)
        rc = self._lib.cimbar_ingest_run_raw(self._h, frames.ctypes.data, n, int(bool(should_preprocess)), int(color_correction), cb, None)
        if rc < 0:
            raise decoder.CimbarHipError(f"cimbar_ingest_run_raw: {rc} {self._lib.cimbar_ingest_last_error(self._h).decode()}")
        return int(rc), chunks, masks

    def run_raw_ptr(self, ptr, n, should_preprocess=False, color_correction=2):
        """frames at a raw host address (e.g. a pinned torch tensor); results discarded, returns good bytes"""
        cb = SINK_FN(lambda user, c, m, first, cnt: 0)
        rc = self._lib.cimbar_ingest_run_raw(self._h, ctypes.c_void_p(ptr), int(n), int(bool(should_preprocess)), int(color_correction), cb, None)
        if rc < 0:
            raise decoder.CimbarHipError(f"cimbar_ingest_run_raw: {rc} {self._lib.cimbar_ingest_last_error(self._h).decode()}")
        return int(rc)

    def png_stats(self):
        out = (ctypes.c_int64 * 4)()
        self._lib.cimbar_ingest_png_stats(self._h, out)
        return {"files": out[0], "refused_by_host_walk": out[1], "refused_by_device": out[2], "bytes_to_device": out[3]}

    def host_decoded(self):
        """device PNG mode, last run_files: files the host threads decoded instead (JPEG, PNGs the kernels do not take)"""
        return int(self._lib.cimbar_ingest_host_decoded(self._h))

    def fallback_overflow(self):
        """device PNG mode, last run_files: readable files dropped because their batch's 32 host-decoded fallback frames were taken"""
        self._lib.cimbar_ingest_fallback_overflow.restype = ctypes.c_int64
        self._lib.cimbar_ingest_fallback_overflow.argtypes = [ctypes.c_void_p]
        return int(self._lib.cimbar_ingest_fallback_overflow(self._h))

    def timings(self):
        out = (ctypes.c_double * 3)()
        self._lib.cimbar_ingest_timings(self._h, out)
        return {"wall_s": out[0], "host_fill_s": out[1], "device_wait_s": out[2]}
