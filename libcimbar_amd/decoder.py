"""ctypes binding of the C ABI in include/cimbar_hip.h (plumbing for tests / bench.py / the multi-GPU driver).

Mirrors the reference's Decoder surface for the hot path (src/lib/encoder/Decoder.h:16-38):
`HipDecoder.decode_fountain(img, sink, should_preprocess, color_correction)` writes the good chunks to `sink.write` in chunk
order and returns the good byte count, exactly what `Decoder::decode_fountain` does through aligned_stream.

There is deliberately no CPU fallback: if the shared library or a gfx950 device is missing, loading/creating raises.
"""
import ctypes
import os

import numpy as np

from . import geometry

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcimbar_hip.so")

MEM_HOST, MEM_DEVICE = 0, 1
TAP_BITPLANE, TAP_SYMBOLS, TAP_COLORS, TAP_DRIFT, TAP_RS_OK, TAP_FLOOD, TAP_CCM, TAP_FLOOD_PATH, TAP_FLOOD_INFO, TAP_FLOOD_VERIFY = range(10)

# every symbol include/cimbar_hip.h declares (tests/test_capi_symbols.py checks the header against this list and the .so)
EXPORTS = (
    "cimbar_hip_create", "cimbar_hip_destroy", "cimbar_hip_bufsize", "cimbar_hip_last_error", "cimbar_hip_decode_frame",
    "cimbar_hip_decode_frame_async", "cimbar_hip_decode_frame_wait",
    "cimbar_hip_decode_batch", "cimbar_hip_reset_ccm", "cimbar_hip_get_ccm", "cimbar_hip_tap", "cimbar_hip_enable_timing",
    "cimbar_hip_stage_times", "cimbar_hip_set_template", "cimbar_hip_encode_batch", "cimbar_hip_decode_plain_batch",
    "cimbar_hip_decode_batch_pipelined", "cimbar_hip_pipeline_wait", "cimbar_hip_pipeline_depth",
    "cimbar_hip_scan_preprocess", "cimbar_hip_deskew_batch", "cimbar_hip_tile_hashes",
    "cimbar_hip_extract_batch", "cimbar_hip_scan_extract_decode_batch", "cimbar_hip_comm_init_all", "cimbar_hip_comm_unique_id",
    "cimbar_hip_comm_init_rank", "cimbar_hip_comm_info", "cimbar_hip_comm_destroy", "cimbar_hip_gather_chunks", "cimbar_hip_pipeline_gather", "cimbar_hip_device", "cimbar_hip_geometry",
    "cimbar_hip_png_scratch_bytes", "cimbar_hip_png_decode_batch", "cimbar_hip_png_decode_batch_v",
    "cimbar_hip_ctx_bufsize", "cimbar_hip_mode_bufsize", "cimbar_hip_set_ccm",
    "cimbar_hip_capture_bytes", "cimbar_hip_scan_preprocess_fmt", "cimbar_hip_deskew_batch_fmt", "cimbar_hip_extract_batch_fmt",
    "cimbar_hip_scan_extract_decode_batch_fmt",
)
PNG_EHEADER, PNG_ESTREAM, PNG_ECODES, PNG_ESIZE, PNG_ECHECK = -30, -31, -32, -33, -34


class PngDesc(ctypes.Structure):
    """cimbar_hip_png_desc"""
    _fields_ = [("zoff", ctypes.c_uint64), ("zlen", ctypes.c_uint32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("color_type", ctypes.c_uint32), ("pal_off", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class CimbarHipError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libcimbar_hip.so and declare the prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("CIMBAR_HIP_LIB") or LIB_PATH   # CIMBAR_HIP_LIB: A/B-test another build of the same ABI
    if not os.path.exists(p):
        raise CimbarHipError(f"{p} not found: build it with `python -m libcimbar_amd.build` (hipcc, gfx950). "
                             "There is no CPU fallback for the decode path.")
    lib = ctypes.CDLL(p)
    vp, i32, u32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int64, ctypes.c_size_t
    lib.cimbar_hip_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
    lib.cimbar_hip_create.restype = i32
    lib.cimbar_hip_destroy.argtypes = [vp]
    lib.cimbar_hip_destroy.restype = None
    lib.cimbar_hip_bufsize.argtypes = []
    lib.cimbar_hip_bufsize.restype = i32
    lib.cimbar_hip_geometry.argtypes = [vp, vp]
    lib.cimbar_hip_geometry.restype = i32
    lib.cimbar_hip_tile_hashes.argtypes = [vp]
    lib.cimbar_hip_tile_hashes.restype = i32
    lib.cimbar_hip_last_error.argtypes = [vp]
    lib.cimbar_hip_last_error.restype = ctypes.c_char_p
    lib.cimbar_hip_decode_frame.argtypes = [vp, vp, u32, u32, sz, i32, i32, vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.cimbar_hip_decode_frame.restype = i32
    lib.cimbar_hip_decode_frame_async.argtypes = [vp, vp, u32, u32, sz, i32, i32, vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.cimbar_hip_decode_frame_async.restype = ctypes.c_longlong
    lib.cimbar_hip_decode_frame_wait.argtypes = [vp, ctypes.c_longlong]
    lib.cimbar_hip_decode_frame_wait.restype = i32
    lib.cimbar_hip_decode_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.cimbar_hip_decode_batch.restype = i64
    lib.cimbar_hip_decode_plain_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.cimbar_hip_decode_plain_batch.restype = i64
    lib.cimbar_hip_decode_batch_pipelined.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.cimbar_hip_decode_batch_pipelined.restype = i32
    lib.cimbar_hip_pipeline_wait.argtypes = [vp, vp, i32]
    lib.cimbar_hip_pipeline_wait.restype = i32
    lib.cimbar_hip_pipeline_depth.argtypes = [vp]
    lib.cimbar_hip_pipeline_depth.restype = i32
    lib.cimbar_hip_scan_preprocess.argtypes = [vp, vp, u32, u32, i32, i32, vp, vp, i32, vp]
    lib.cimbar_hip_scan_preprocess.restype = i32
    lib.cimbar_hip_deskew_batch.argtypes = [vp, vp, u32, u32, i32, i32, vp, vp, i32, vp]
    lib.cimbar_hip_deskew_batch.restype = i32
    lib.cimbar_hip_extract_batch.argtypes = [vp, vp, u32, u32, i32, i32, vp, vp, vp, i32, vp]
    lib.cimbar_hip_extract_batch.restype = i32
    lib.cimbar_hip_scan_extract_decode_batch.argtypes = [vp, vp, u32, u32, i32, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.cimbar_hip_scan_extract_decode_batch.restype = i64
    lib.cimbar_hip_capture_bytes.argtypes = [u32, u32, i32]
    lib.cimbar_hip_capture_bytes.restype = sz
    lib.cimbar_hip_scan_preprocess_fmt.argtypes = [vp, vp, u32, u32, i32, i32, i32, vp, vp, i32, vp]
    lib.cimbar_hip_scan_preprocess_fmt.restype = i32
    lib.cimbar_hip_deskew_batch_fmt.argtypes = [vp, vp, u32, u32, i32, i32, i32, vp, vp, i32, vp]
    lib.cimbar_hip_deskew_batch_fmt.restype = i32
    lib.cimbar_hip_extract_batch_fmt.argtypes = [vp, vp, u32, u32, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.cimbar_hip_extract_batch_fmt.restype = i32
    lib.cimbar_hip_scan_extract_decode_batch_fmt.argtypes = [vp, vp, u32, u32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.cimbar_hip_scan_extract_decode_batch_fmt.restype = i64
    lib.cimbar_hip_comm_init_all.argtypes = [i32, vp, ctypes.POINTER(vp)]
    lib.cimbar_hip_comm_init_all.restype = i32
    lib.cimbar_hip_comm_unique_id.argtypes = [vp]
    lib.cimbar_hip_comm_unique_id.restype = i32
    lib.cimbar_hip_comm_init_rank.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp)]
    lib.cimbar_hip_comm_init_rank.restype = i32
    lib.cimbar_hip_comm_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.cimbar_hip_comm_info.restype = i32
    lib.cimbar_hip_comm_destroy.argtypes = [vp]
    lib.cimbar_hip_comm_destroy.restype = None
    lib.cimbar_hip_gather_chunks.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.cimbar_hip_gather_chunks.restype = i32
    lib.cimbar_hip_pipeline_gather.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp]
    lib.cimbar_hip_pipeline_gather.restype = i32
    lib.cimbar_hip_device.argtypes = [vp]
    lib.cimbar_hip_device.restype = i32
    lib.cimbar_hip_reset_ccm.argtypes = [vp]
    lib.cimbar_hip_reset_ccm.restype = i32
    lib.cimbar_hip_get_ccm.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.cimbar_hip_get_ccm.restype = i32
    lib.cimbar_hip_set_ccm.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.cimbar_hip_set_ccm.restype = i32
    lib.cimbar_hip_mode_bufsize.argtypes = [i32]
    lib.cimbar_hip_mode_bufsize.restype = i32
    lib.cimbar_hip_ctx_bufsize.argtypes = [vp]
    lib.cimbar_hip_ctx_bufsize.restype = i32
    lib.cimbar_hip_tap.argtypes = [vp, i32, vp, sz]
    lib.cimbar_hip_tap.restype = i64
    lib.cimbar_hip_enable_timing.argtypes = [vp, i32]
    lib.cimbar_hip_enable_timing.restype = i32
    lib.cimbar_hip_stage_times.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), i32]
    lib.cimbar_hip_stage_times.restype = i32
    lib.cimbar_hip_set_template.argtypes = [vp, vp, i32]
    lib.cimbar_hip_set_template.restype = i32
    lib.cimbar_hip_encode_batch.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    lib.cimbar_hip_encode_batch.restype = i32
    lib.cimbar_hip_png_scratch_bytes.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
    lib.cimbar_hip_png_scratch_bytes.restype = sz
    lib.cimbar_hip_png_decode_batch.argtypes = [i32, vp, sz, vp, i32, vp, sz, vp, sz, vp, vp]
    lib.cimbar_hip_png_decode_batch.restype = i32
    lib.cimbar_hip_png_decode_batch_v.argtypes = [i32, vp, sz, vp, i32, vp, sz, vp, sz, vp, i32, vp]
    lib.cimbar_hip_png_decode_batch_v.restype = i32
    if path is None:
        _lib = lib
    return lib


def png_split(png):
    """PNG bytes -> (width, height, colour type, bit depth, interlace, zlib stream = the concatenated IDAT payloads, palette bytes or None):
    the chunk walk the ingest library's device mode does on the host"""
    import struct
    if png[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, z, pal, hdr = 8, [], None, None
    while pos + 12 <= len(png):
        n, tag = struct.unpack(">I4s", png[pos:pos + 8])
        body = png[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            pal = bytes(body)
        elif tag == b"IDAT":
            z.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _c, _f, interlace = hdr
    return w, h, ctype, depth, interlace, b"".join(z), pal


def png_decode_batch_device(pngs, device=0, variant=0):
    """PNG files (bytes) -> (list of (h, w, 3) uint8 arrays or None, status array): cimbar_hip_png_decode_batch on torch device buffers.
    Test / bench plumbing: packs the streams the way libcimbar_ingest.so's device mode does."""
    import torch
    lib = load_library()
    n = len(pngs)
    desc = (PngDesc * n)()
    blob = bytearray()
    dims = []
    for i, png in enumerate(pngs):
        w, h, ctype, depth, interlace, z, pal = png_split(png)
        if depth != 8 or interlace:
            raise ValueError("the device decoder takes 8-bit non-interlaced PNGs")
        while len(blob) % 16:
            blob.append(0)
        desc[i].zoff, desc[i].zlen, desc[i].width, desc[i].height, desc[i].color_type = len(blob), len(z), w, h, ctype
        blob += z
        if ctype == 3:
            while len(blob) % 16:
                blob.append(0)
            desc[i].pal_off = len(blob)
            blob += (pal or b"") + bytes(768 - len(pal or b""))
        dims.append((w, h, ctype))
    while len(blob) % 16:
        blob.append(0)
    dev = torch.device("cuda", device)
    d_z = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).to(dev)
    d_desc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
    sstride = max(int(lib.cimbar_hip_png_scratch_bytes(w, h, ct)) for w, h, ct in dims)
    rstride = (max(w * h * 3 for w, h, _ in dims) + 15) & ~15
    d_scratch = torch.empty(n * sstride, dtype=torch.uint8, device=dev)
    d_rgb = torch.zeros(n * rstride, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), 12345, dtype=torch.int32, device=dev)
    # variant: cimbar_hip_png_decode_batch_v's (0 = chosen by n, 1 = one stream per wavefront, 4 = "many in flight": the device chooses the kernel)
    lib.cimbar_hip_png_decode_batch_v.restype = ctypes.c_int
    rc = lib.cimbar_hip_png_decode_batch_v(device, ctypes.c_void_p(d_z.data_ptr()), ctypes.c_size_t(d_z.numel()), ctypes.c_void_p(d_desc.data_ptr()), n,
                                           ctypes.c_void_p(d_scratch.data_ptr()), ctypes.c_size_t(sstride), ctypes.c_void_p(d_rgb.data_ptr()), ctypes.c_size_t(rstride),
                                           ctypes.c_void_p(d_status.data_ptr()), int(variant), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise CimbarHipError(f"cimbar_hip_png_decode_batch: {rc}")
    torch.cuda.synchronize(dev)
    status = d_status.cpu().numpy()
    rgb = d_rgb.cpu().numpy()
    out = [rgb[i * rstride:i * rstride + w * h * 3].reshape(h, w, 3).copy() if status[i] == 0 else None for i, (w, h, _) in enumerate(dims)]
    return out, status


def tile_hashes():
    """the 16 tile hashes the library computes at create time (host arithmetic, no device needed)"""
    out = np.zeros(16, dtype=np.uint64)
    load_library().cimbar_hip_tile_hashes(out.ctypes.data)
    return out


def comm_unique_id():
    """128 bytes naming a new RCCL communicator (rank 0 makes it, every rank passes it to HipDecoder.comm_init_rank)"""
    buf = (ctypes.c_uint8 * 128)()
    rc = load_library().cimbar_hip_comm_unique_id(buf)
    if rc != 0:
        raise CimbarHipError(f"cimbar_hip_comm_unique_id failed: {rc} (RCCL not available?)")
    return bytes(buf)


def comm_info(comm):
    """(nranks, rank) of a communicator as RCCL itself reports them (ncclCommCount / ncclCommUserRank)"""
    n, r = ctypes.c_int32(0), ctypes.c_int32(-1)
    rc = load_library().cimbar_hip_comm_info(comm, ctypes.byref(n), ctypes.byref(r))
    if rc != 0:
        raise CimbarHipError(f"cimbar_hip_comm_info failed: {rc}")
    return int(n.value), int(r.value)


def comm_destroy(comm):
    load_library().cimbar_hip_comm_destroy(comm)


_ERR = {-1: "EINVAL", -2: "EDIM", -3: "ENODEVICE", -4: "EHIP", -5: "ENOMEM"}


class HipDecoder:
    """One decode context on one GPU (== one reference `Decoder` + its thread_local colour-correction state)."""

    def __init__(self, device=0, mode=68, lib_path=None):
        self._lib = load_library(lib_path)   # lib_path: another build of the same ABI (tests: the spill-path variant)
        self._ctx = ctypes.c_void_p()
        rc = self._lib.cimbar_hip_create(int(device), int(mode), ctypes.byref(self._ctx))
        if rc != 0:
            self._ctx = ctypes.c_void_p()
            raise CimbarHipError(f"cimbar_hip_create(device={device}, mode={mode}) failed: {_ERR.get(rc, rc)} "
                                 "(a gfx950 GPU is required; there is no CPU fallback)")
        self.device = device
        g = (ctypes.c_int32 * 12)()
        self._check(self._lib.cimbar_hip_geometry(self._ctx, g), "cimbar_hip_geometry")
        self.geo = geometry.for_mode(g[0])
        if (self.geo.IMG_W, self.geo.IMG_H, self.geo.NCELLS, self.geo.CHUNKS_PER_FRAME, self.geo.CHUNK, self.geo.BLOCKS, self.geo.RS_BLOCK,
                self.geo.RS_PARITY, self.geo.DIM_X, self.geo.DIM_Y, self.geo.OFFSET) != tuple(g[1:12]):
            raise CimbarHipError(f"library geometry {list(g)} does not match libcimbar_amd.geometry for mode {g[0]}")

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.cimbar_hip_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            msg = self._lib.cimbar_hip_last_error(self._ctx).decode("utf-8", "replace")
            raise CimbarHipError(f"{what}: {_ERR.get(int(rc), rc)} {msg}")
        return rc

    # ------------------------------------------------------------------ host-memory entry points
    def decode_frame(self, rgb, should_preprocess=False, color_correction=2):
        """rgb: (1024,1024,3) uint8 numpy array. Returns (good_bytes, chunks (12,625) uint8, mask int)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise CimbarHipError("decode_frame: expected an HxWx3 uint8 image")
        chunks = np.zeros((self.geo.CHUNKS_PER_FRAME, self.geo.CHUNK), dtype=np.uint8)
        mask = ctypes.c_uint32(0)
        rc = self._lib.cimbar_hip_decode_frame(self._ctx, rgb.ctypes.data, rgb.shape[1], rgb.shape[0], rgb.strides[0],
                                               int(bool(should_preprocess)), int(color_correction), chunks.ctypes.data,
                                               ctypes.byref(mask))
        self._check(rc, "cimbar_hip_decode_frame")
        return rc, chunks, mask.value

    def decode_frame_async(self, rgb, should_preprocess=False, color_correction=2):
        """Starts one frame (cimbar_hip_decode_frame_async) and returns a ticket for decode_frame_wait; up to pipeline_depth frames in flight, the
        next frame's host-to-device copy running beside this one's kernels. `rgb` must stay alive and untouched until the wait when it is
        page-locked memory (the array is kept referenced here either way)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise CimbarHipError("decode_frame_async: expected an HxWx3 uint8 image")
        chunks = np.zeros((self.geo.CHUNKS_PER_FRAME, self.geo.CHUNK), dtype=np.uint8)
        mask = ctypes.c_uint32(0)
        t = self._lib.cimbar_hip_decode_frame_async(self._ctx, rgb.ctypes.data, rgb.shape[1], rgb.shape[0], rgb.strides[0],
                                                    int(bool(should_preprocess)), int(color_correction), chunks.ctypes.data, ctypes.byref(mask))
        self._check(t, "cimbar_hip_decode_frame_async")
        if not hasattr(self, "_frames_in_flight"):
            self._frames_in_flight = {}
        self._frames_in_flight[int(t)] = (rgb, chunks, mask)
        for old in [k for k in self._frames_in_flight if k < int(t) - 16]:      # (tickets nobody waited for)
            del self._frames_in_flight[old]
        return int(t)

    def decode_frame_wait(self, ticket):
        """Blocks until the frame is complete. Returns (good_bytes, chunks (12,625) uint8, mask int) like decode_frame."""
        rc = self._lib.cimbar_hip_decode_frame_wait(self._ctx, int(ticket))
        self._check(rc, "cimbar_hip_decode_frame_wait")
        _rgb, chunks, mask = self._frames_in_flight.pop(int(ticket))
        return rc, chunks, mask.value

    def decode_batch(self, frames, should_preprocess=False, color_correction=2):
        """frames: (n,1024,1024,3) uint8 numpy. Returns (total_good_bytes, chunks (n,12,625), masks (n,) uint32)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        if frames.shape[1:] != self.geo.FRAME_SHAPE:
            raise CimbarHipError("decode_batch: frames must be (n,1024,1024,3) uint8")
        chunks = np.zeros((n, self.geo.CHUNKS_PER_FRAME, self.geo.CHUNK), dtype=np.uint8)
        masks = np.zeros(n, dtype=np.uint32)
        rc = self._lib.cimbar_hip_decode_batch(self._ctx, frames.ctypes.data, n, MEM_HOST, int(bool(should_preprocess)),
                                               int(color_correction), chunks.ctypes.data, masks.ctypes.data, MEM_HOST, None)
        self._check(rc, "cimbar_hip_decode_batch")
        return int(rc), chunks, masks

    def decode_plain_batch(self, frames, should_preprocess=False, color_correction=2):
        """Decoder::decode (the --no-fountain path) for frames (n,1024,1024,3) uint8 numpy. Returns (bytes_written,
        data (n,7500) uint8 with failed RS blocks zeroed, block_ok (n,60) uint8)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        if frames.shape[1:] != self.geo.FRAME_SHAPE:
            raise CimbarHipError("decode_plain_batch: frames must be (n,1024,1024,3) uint8")
        data = np.zeros((n, self.geo.FRAME_BYTES), dtype=np.uint8)
        ok = np.zeros((n, self.geo.BLOCKS), dtype=np.uint8)
        rc = self._lib.cimbar_hip_decode_plain_batch(self._ctx, frames.ctypes.data, n, MEM_HOST, int(bool(should_preprocess)),
                                                     int(color_correction), data.ctypes.data, ok.ctypes.data, MEM_HOST, None)
        self._check(rc, "cimbar_hip_decode_plain_batch")
        return int(rc), data, ok

    # ------------------------------------------------------------------ device-memory entry point (torch tensors as raw memory)
    def decode_batch_device(self, frames_ptr, n, chunks_ptr, masks_ptr, should_preprocess=False, color_correction=2, stream=None):
        """Enqueue a batch whose input and outputs are device pointers (ints) on HIP stream handle `stream` (None / 0 = the null
        stream, i.e. torch's default stream). Asynchronous: the caller synchronises that stream."""
        rc = self._lib.cimbar_hip_decode_batch(self._ctx, ctypes.c_void_p(frames_ptr), int(n), MEM_DEVICE,
                                               int(bool(should_preprocess)), int(color_correction), ctypes.c_void_p(chunks_ptr),
                                               ctypes.c_void_p(masks_ptr), MEM_DEVICE,
                                               ctypes.c_void_p(stream) if stream else None)
        self._check(rc, "cimbar_hip_decode_batch(device)")
        return int(rc)

    def decode_batch_pipelined(self, frames_ptr, n, chunks_ptr, masks_ptr, should_preprocess=False, color_correction=2, stream=None):
        """Like decode_batch_device, but only the threshold pass runs on `stream`: the rest of the batch overlaps the next batch's
        threshold pass. Outputs are valid once a later pipeline_wait() on a stream has been reached; at most pipeline_depth batches in
        flight."""
        rc = self._lib.cimbar_hip_decode_batch_pipelined(self._ctx, ctypes.c_void_p(frames_ptr), int(n), int(bool(should_preprocess)),
                                                         int(color_correction), ctypes.c_void_p(chunks_ptr), ctypes.c_void_p(masks_ptr),
                                                         ctypes.c_void_p(stream) if stream else None)
        self._check(rc, "cimbar_hip_decode_batch_pipelined")

    def pipeline_wait(self, stream=None, keep_newest=0):
        """`stream` waits for the pipelined batches issued so far except the `keep_newest` most recent ones."""
        self._check(self._lib.cimbar_hip_pipeline_wait(self._ctx, ctypes.c_void_p(stream) if stream else None, int(keep_newest)),
                    "cimbar_hip_pipeline_wait")

    @property
    def pipeline_depth(self):
        return int(self._lib.cimbar_hip_pipeline_depth(self._ctx))

    # ------------------------------------------------------------------ the stage in front: Scanner's image preparation, Deskewer
    def _captures(self, captures, size, fmt):
        """RGB8 captures as an (n,h,w,3) array, or -- with size=(w,h) and the C ABI's `fmt` (3 RGB, 4 RGBA, 12 NV12, 420) -- n captures of
        cimbar_hip_capture_bytes(w, h, fmt) bytes each as an (n, bytes) array. Returns (array, n, w, h, fmt)."""
        captures = np.ascontiguousarray(captures, dtype=np.uint8)
        if size is None:
            fmt = int(fmt) if int(fmt) > 0 else 3          # (<= 0 means RGB to the C ABI as well, cimbar_recv_js.cpp:150-151)
            if fmt == 4 and captures.ndim == 4 and captures.shape[3] == 4:
                n, h, w = captures.shape[:3]
                return captures, n, w, h, 4
            if fmt != 3:
                raise CimbarHipError(f"captures in format {fmt} need size=(w, h): the array's shape does not say what the frame is")
            if captures.ndim != 4 or captures.shape[3] != 3:
                raise CimbarHipError(f"RGB captures are an (n, h, w, 3) array, got {captures.shape}")
            n, h, w = captures.shape[:3]
            return captures, n, w, h, 3
        w, h = size
        per = int(self._lib.cimbar_hip_capture_bytes(w, h, int(fmt)))
        if per == 0 or captures.size % per:
            raise CimbarHipError(f"captures of {w}x{h} in format {fmt}: {captures.size} bytes is no multiple of {per}")
        return captures, captures.size // per, w, h, int(fmt)

    def scan_preprocess(self, captures, size=None, fmt=3):
        """captures -> (binary (n,h,w) uint8 of 0/255, thresholds (n,) int32): Scanner::preprocess_image."""
        captures, n, w, h, fmt = self._captures(captures, size, fmt)
        out = np.zeros((n, h, w), dtype=np.uint8)
        thr = np.zeros(n, dtype=np.int32)
        self._check(self._lib.cimbar_hip_scan_preprocess_fmt(self._ctx, captures.ctypes.data, w, h, fmt, n, MEM_HOST, out.ctypes.data, thr.ctypes.data,
                                                             MEM_HOST, None), "cimbar_hip_scan_preprocess_fmt")
        return out, thr

    def deskew_batch(self, captures, corners, size=None, fmt=3):
        """captures, corners (n,8) float32 (tl, tr, bl, br as x,y) -> frames (n,1024,1024,3): Deskewer::deskew."""
        captures, n, w, h, fmt = self._captures(captures, size, fmt)
        corners = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((n, *self.geo.FRAME_SHAPE), dtype=np.uint8)
        self._check(self._lib.cimbar_hip_deskew_batch_fmt(self._ctx, captures.ctypes.data, w, h, fmt, n, MEM_HOST, corners.ctypes.data, out.ctypes.data,
                                                          MEM_HOST, None), "cimbar_hip_deskew_batch_fmt")
        return out

    def deskew_batch_device(self, captures_ptr, w, h, n, corners, frames_ptr, stream=None, fmt=3):
        """device captures -> device frames (what cimbar_hip_decode_batch takes next); corners stay a host array."""
        corners = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 8)
        self._check(self._lib.cimbar_hip_deskew_batch_fmt(self._ctx, ctypes.c_void_p(captures_ptr), int(w), int(h), int(fmt), int(n), MEM_DEVICE,
                                                          corners.ctypes.data, ctypes.c_void_p(frames_ptr), MEM_DEVICE,
                                                          ctypes.c_void_p(stream) if stream else None), "cimbar_hip_deskew_batch_fmt(device)")

    def extract_batch(self, captures, size=None, fmt=3):
        """Extractor::extract for captures -> (status (n,) int32, corners (n,8) float32, frames (n,1024,1024,3))"""
        captures, n, w, h, fmt = self._captures(captures, size, fmt)
        frames = np.zeros((n, *self.geo.FRAME_SHAPE), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        corners = np.zeros((n, 8), dtype=np.float32)
        self._check(self._lib.cimbar_hip_extract_batch_fmt(self._ctx, captures.ctypes.data, w, h, fmt, n, MEM_HOST, frames.ctypes.data, status.ctypes.data,
                                                           corners.ctypes.data, MEM_HOST, None), "cimbar_hip_extract_batch_fmt")
        return status, corners, frames

    def scan_extract_decode_batch(self, captures, preprocess=-1, color_correction=2, size=None, fmt=3):
        """cimbard_scan_extract_decode for n captures -> (good_bytes, chunks (n,12,625), masks (n,), status (n,))"""
        captures, n, w, h, fmt = self._captures(captures, size, fmt)
        chunks = np.zeros((n, self.geo.CHUNKS_PER_FRAME, self.geo.CHUNK), dtype=np.uint8)
        masks = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.int32)
        rc = self._check(self._lib.cimbar_hip_scan_extract_decode_batch_fmt(self._ctx, captures.ctypes.data, w, h, fmt, n, MEM_HOST, int(preprocess),
                                                                             int(color_correction), chunks.ctypes.data, masks.ctypes.data,
                                                                             status.ctypes.data, MEM_HOST, None), "cimbar_hip_scan_extract_decode_batch_fmt")
        return int(rc), chunks, masks, status

    def scan_extract_decode_device(self, captures_ptr, w, h, n, chunks_ptr, masks_ptr, status_ptr=None, preprocess=-1, color_correction=2, stream=None, fmt=3):
        """device captures in, device chunks / masks / status out; asynchronous on `stream`"""
        self._check(self._lib.cimbar_hip_scan_extract_decode_batch_fmt(self._ctx, ctypes.c_void_p(captures_ptr), int(w), int(h), int(fmt), int(n), MEM_DEVICE,
                                                                        int(preprocess), int(color_correction), ctypes.c_void_p(chunks_ptr),
                                                                        ctypes.c_void_p(masks_ptr), ctypes.c_void_p(status_ptr) if status_ptr else None,
                                                                        MEM_DEVICE, ctypes.c_void_p(stream) if stream else None),
                    "cimbar_hip_scan_extract_decode_batch_fmt(device)")

    # ------------------------------------------------------------------ the reference's operator surface
    def decode_fountain(self, img, ostream, should_preprocess=False, color_correction=2):
        """Decoder::decode_fountain (Decoder.h:171-189): good chunks go to ostream.write(bytes) in chunk order; returns good bytes.
        Like the reference, a sink whose chunk_size() is not 625 gets nothing written but the byte count is still returned."""
        good, chunks, mask = self.decode_frame(img, should_preprocess, color_correction)
        feed = True
        if hasattr(ostream, "chunk_size") and ostream.chunk_size() != self.geo.CHUNK:
            feed = False
        if feed:
            for j in range(self.geo.CHUNKS_PER_FRAME):
                if mask & (1 << j):
                    ostream.write(chunks[j].tobytes())
        return good

    # ------------------------------------------------------------------ encode half (frame synthesiser)
    def _ensure_template(self):
        if getattr(self, "_have_template", False):
            return
        path = os.path.join(_HERE, "data", self.geo.TEMPLATE)
        z = np.load(path)
        t = np.zeros(self.geo.FRAME_RGB_BYTES, dtype=np.uint8)
        t[z["idx"]] = z["val"]
        self._check(self._lib.cimbar_hip_set_template(self._ctx, t.ctypes.data, MEM_HOST), "cimbar_hip_set_template")
        self._have_template = True

    def encode_batch(self, payload):
        """payload (n,7500) uint8 numpy -> frames (n,1024,1024,3) uint8 numpy (Encoder::encode_next for n frames)."""
        self._ensure_template()
        payload = np.ascontiguousarray(payload, dtype=np.uint8).reshape(-1, self.geo.FRAME_BYTES)
        n = payload.shape[0]
        out = np.empty((n, *self.geo.FRAME_SHAPE), dtype=np.uint8)
        self._check(self._lib.cimbar_hip_encode_batch(self._ctx, payload.ctypes.data, n, MEM_HOST, out.ctypes.data, MEM_HOST, None),
                    "cimbar_hip_encode_batch")
        return out

    def encode_batch_device(self, payload_ptr, n, rgb_ptr, stream=None):
        """Device pointers in and out; asynchronous on `stream` (None / 0 = the null stream)."""
        self._ensure_template()
        self._check(self._lib.cimbar_hip_encode_batch(self._ctx, ctypes.c_void_p(payload_ptr), int(n), MEM_DEVICE, ctypes.c_void_p(rgb_ptr),
                                                      MEM_DEVICE, ctypes.c_void_p(stream) if stream else None),
                    "cimbar_hip_encode_batch(device)")

    # ------------------------------------------------------------------ multi-GPU exchange through the library's own RCCL binding
    def comm_init_rank(self, uid, nranks, rank):
        """join the communicator named by the 128-byte `uid` (bytes from comm_unique_id() on rank 0); returns an opaque handle"""
        comm = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(uid))
        self._check(self._lib.cimbar_hip_comm_init_rank(buf, int(nranks), int(rank), int(self.device), ctypes.byref(comm)), "cimbar_hip_comm_init_rank")
        return comm

    def pipeline_gather(self, comm, root, chunks_ptr, masks_ptr, n, all_chunks_ptr, all_masks_ptr):
        """cimbar_hip_pipeline_gather: the exchange of the pipelined batch issued last, on that batch's own stream (pipeline_wait then covers it)."""
        self._check(self._lib.cimbar_hip_pipeline_gather(self._ctx, comm, int(root), ctypes.c_void_p(chunks_ptr), ctypes.c_void_p(masks_ptr), int(n),
                                                         ctypes.c_void_p(all_chunks_ptr) if all_chunks_ptr else None,
                                                         ctypes.c_void_p(all_masks_ptr) if all_masks_ptr else None), "cimbar_hip_pipeline_gather")

    def gather_chunks(self, comm, root, chunks_ptr, masks_ptr, n, all_chunks_ptr, all_masks_ptr, stream=None):
        self._check(self._lib.cimbar_hip_gather_chunks(self._ctx, comm, int(root), ctypes.c_void_p(chunks_ptr), ctypes.c_void_p(masks_ptr), int(n),
                                                       ctypes.c_void_p(all_chunks_ptr) if all_chunks_ptr else None,
                                                       ctypes.c_void_p(all_masks_ptr) if all_masks_ptr else None,
                                                       ctypes.c_void_p(stream) if stream else None), "cimbar_hip_gather_chunks")

    # ------------------------------------------------------------------ state / taps / timing
    def reset_ccm(self):
        self._check(self._lib.cimbar_hip_reset_ccm(self._ctx), "cimbar_hip_reset_ccm")

    def get_ccm(self):
        out = (ctypes.c_float * 9)()
        rc = self._check(self._lib.cimbar_hip_get_ccm(self._ctx, out), "cimbar_hip_get_ccm")
        return bool(rc), np.array(list(out), dtype=np.float32).reshape(3, 3)

    def set_ccm(self, m):
        """CimbDecoder::update_color_correction: the carried matrix becomes `m` (3x3) and is active from the next frame on"""
        arr = (ctypes.c_float * 9)(*[float(x) for x in np.asarray(m, dtype=np.float32).reshape(-1)])
        self._check(self._lib.cimbar_hip_set_ccm(self._ctx, arr), "cimbar_hip_set_ccm")

    def bufsize(self):
        """cimbard_get_bufsize() of this context's configuration"""
        return self._check(self._lib.cimbar_hip_ctx_bufsize(self._ctx), "cimbar_hip_ctx_bufsize")

    def tap(self, what, n):
        shapes = {
            TAP_BITPLANE: ((n, self.geo.IMG_W * self.geo.IMG_H // 8), np.uint8), TAP_SYMBOLS: ((n, self.geo.NCELLS), np.uint8),
            TAP_COLORS: ((n, self.geo.NCELLS), np.uint8), TAP_DRIFT: ((n, self.geo.NCELLS, 2), np.int8),
            TAP_RS_OK: ((n, self.geo.BLOCKS), np.uint8), TAP_FLOOD: ((n,), np.uint8), TAP_CCM: ((n, 10), np.float32), TAP_FLOOD_PATH: ((n,), np.uint8), TAP_FLOOD_INFO: ((n,), np.uint32), TAP_FLOOD_VERIFY: ((n,), np.uint32),
        }
        shape, dt = shapes[what]
        out = np.zeros(shape, dtype=dt)
        self._check(self._lib.cimbar_hip_tap(self._ctx, what, out.ctypes.data, out.nbytes), "cimbar_hip_tap")
        return out

    def enable_timing(self, on=True):
        self._lib.cimbar_hip_enable_timing(self._ctx, int(bool(on)))

    def stage_times(self):
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_float * 16)()
        k = self._check(self._lib.cimbar_hip_stage_times(self._ctx, names, ms, 16), "cimbar_hip_stage_times")
        return {names[i].decode(): float(ms[i]) for i in range(k)}
