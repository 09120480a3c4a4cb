"""GPU: the loop of the reference's receiver (/root/reference/src/exe/cimbar_recv2/recv2.cpp:109-151: cimbard_configure_decode, cimbard_get_bufsize,
then per camera frame cimbard_scan_extract_decode -> cimbard_fountain_decode -> on completion cimbard_get_filesize / _get_filename /
_decompress_read) run twice on the same captures: against the untouched reference build (oracle/_ref/libcimbar_ref.so, which holds the reference's own
cimbar_recv_js.cpp) and against the reference's file with its three decode-path functions replaced by the product's (oracle/_ref/
libcimbar_recv_full.so = libcimbar_amd/host/cimbar_recv_c.cpp + the rest of cimbar_recv_js.cpp, oracle/Makefile `recvfull`). Return values, packed
chunk buffers, the completed file's id, name and contents must be equal, in all four capture formats and across a mode change."""
import ctypes
import os

import numpy as np
import pytest

from libcimbar_amd import framegen
from oracle import pyref
from oracle.pyref import P
from tests import capture_formats as CF
from tests import frames as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "oracle", "_ref", "libcimbar_recv_full.so")
QUADS = [((500, 40), (1480, 70), (470, 1030), (1500, 1000)), ((420, 60), (1400, 30), (450, 1010), (1430, 1040)),
         ((560, 90), (1450, 50), (540, 980), (1490, 1020)), ((480, 30), (1500, 60), (500, 1050), (1470, 1000))]


def api(lib):
    lib.cimbard_fountain_decode.restype = ctypes.c_int64
    lib.cimbard_get_filesize.restype = ctypes.c_uint
    lib.cimbard_get_filesize.argtypes = [ctypes.c_uint32]
    lib.cimbard_get_filename.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint]
    lib.cimbard_decompress_read.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint]
    return lib


def receiver_loop(lib, mode, captures, w, h):
    """recv2.cpp:109-151 without the camera and the window; returns what it saw at every step"""
    seen = []
    assert lib.cimbard_configure_decode(mode) == 0
    chunk = pyref.GEOMETRY[mode][3]
    bufspace = np.zeros(lib.cimbard_get_bufsize(), np.uint8)
    seen.append(("bufsize", bufspace.size))
    for img, fmt in captures:
        nbytes = lib.cimbard_scan_extract_decode(P(img), w, h, fmt, P(bufspace), bufspace.size)
        seen.append(("sed", nbytes, bufspace[: max(nbytes, 0)].tobytes()))
        if nbytes <= 0 or nbytes % chunk != 0:
            continue
        res = lib.cimbard_fountain_decode(P(bufspace), nbytes)
        seen.append(("fountain", res))
        if res > 0:
            fid = ctypes.c_uint32(res & 0xFFFFFFFF)
            size = lib.cimbard_get_filesize(fid)
            name = ctypes.create_string_buffer(255)
            fnsize = lib.cimbard_get_filename(fid, name, 255)
            data = bytearray()
            piece = np.zeros(lib.cimbard_get_decompress_bufsize(), np.uint8)
            while True:
                got = lib.cimbard_decompress_read(fid, P(piece), piece.size)
                if got <= 0:
                    break
                data += piece[:got].tobytes()
            seen.append(("file", res, size, name.raw[: max(fnsize, 0)], bytes(data)))
            break
    return seen


@pytest.mark.parametrize("mode", [68, 67])
def test_recv2_loop_reference_against_product(hip_decoder, ref, mode):
    if not os.path.exists(FULL):
        pytest.skip("oracle/_ref/libcimbar_recv_full.so not built (needs /root/reference at build time)")
    full = api(ctypes.CDLL(FULL))
    api(ref)
    g = pyref.GEOMETRY[mode]
    w, h = 1920, 1080
    data = np.random.default_rng(77 + mode).integers(0, 64, 21000, dtype=np.uint8)          # compressible: zstd makes a few frames of it
    nframes = 8
    frames = np.zeros((nframes, g[1], g[0], 3), np.uint8)
    with pyref.ref_mode(mode):
        assert ref.ref_encode_fountain_z(P(data), data.size, 109, 6, b"/tmp/some dir/recv_me.bin", 0, nframes, P(frames)) == nframes
    captures = []
    for k in range(nframes):
        fmt = CF.FORMATS[k % 4]
        cam = F.camera_frame(frames[k], width=w, height=h, quad=QUADS[k % len(QUADS)], background=40 + 30 * (k % 3), blur=0.0 if k % 2 else 0.6)
        captures.append((np.ascontiguousarray(CF.rgb_to_format(cam, fmt)), fmt))
    blank = np.zeros(CF.capture_bytes(w, h, 3), np.uint8)
    captures.insert(2, (blank, 3))                                                          # a capture without a frame: -3 on both sides
    ref.ref_reset_ccm()
    want = receiver_loop(ref, mode, captures, w, h)
    got = receiver_loop(full, mode, captures, w, h)
    ref.cimbard_configure_decode(68)
    full.cimbard_configure_decode(68)
    ref.ref_configure(68)
    assert [s[:2] for s in got] == [s[:2] for s in want]
    assert got == want
    assert ("sed", -3, b"") in want and want[-1][0] == "file"
    assert want[-1][3] == b"recv_me.bin" and want[-1][4] == data.tobytes()
