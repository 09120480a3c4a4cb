"""GPU edge cases of the C ABI: padded rows, ragged batch sizes, capacity growth, degenerate frames, bad arguments.
The reference's own tests exercise these through CimbReader/Decoder with odd inputs (DecoderTest.cpp, CimbReaderTest.cpp)."""
import ctypes

import numpy as np
import pytest

from libcimbar_amd import decoder as D
from libcimbar_amd import modeb
from oracle import pyref
from tests import frames as F

pytestmark = pytest.mark.gpu


def test_padded_row_stride(hip_decoder, synth):
    """cv::Mat rows need not be dense (Mat::step): the strided single-frame entry point must ignore the padding."""
    payload, frames = F.clean_frames(synth, 1, seed=77)
    fr = F.add_noise(frames[0], 12, 5)
    stride = 1024 * 3 + 64
    padded = np.full((1024, stride), 0xA5, np.uint8)
    padded[:, :1024 * 3] = fr.reshape(1024, -1)
    lib = D.load_library()
    chunks = np.zeros((12, 625), np.uint8)
    mask = ctypes.c_uint32(0)
    hip_decoder.reset_ccm()
    rc = lib.cimbar_hip_decode_frame(hip_decoder._ctx, padded.ctypes.data, 1024, 1024, stride, 0, 2, chunks.ctypes.data, ctypes.byref(mask))
    hip_decoder.reset_ccm()
    good, want, wmask = hip_decoder.decode_frame(fr)
    assert rc == good and mask.value == wmask and (chunks == want).all()
    assert good == 7500 and (chunks.reshape(-1) == payload[0]).all()


@pytest.mark.parametrize("sizes", [(1, 3, 5, 7, 2), (9, 1, 9)])
def test_ragged_batch_sizes_and_capacity_changes(hip_decoder, synth, sizes):
    """any n (not a multiple of anything), growing and shrinking between calls: per-frame results do not depend on the batch."""
    nmax = max(sizes)
    payload, frames = F.clean_frames(synth, nmax, seed=5)
    frames = [F.add_noise(f, 10 + 3 * k, k) for k, f in enumerate(frames)]
    frames[nmax // 2] = F.shift(frames[nmax // 2], 2, -1)
    frames = np.ascontiguousarray(np.stack(frames))
    hip_decoder.reset_ccm()
    _, ref_chunks, ref_masks = hip_decoder.decode_batch(frames, color_correction=0)
    ref_sym = hip_decoder.tap(D.TAP_SYMBOLS, nmax).copy()
    for n in sizes:
        total, chunks, masks = hip_decoder.decode_batch(frames[:n], color_correction=0)
        assert (masks == ref_masks[:n]).all() and (chunks == ref_chunks[:n]).all()
        assert total == 625 * sum(bin(int(m)).count("1") for m in masks)
        assert (hip_decoder.tap(D.TAP_SYMBOLS, n) == ref_sym[:n]).all()


def test_degenerate_frames_match_oracle(hip_decoder):
    """black, white, and pure-noise frames: nothing crashes and every stage still equals the oracle (a black frame is all-zero RS
    codewords, so the reference delivers some all-zero chunks from it; so must we)
    (noise drives every cell through the flood-order pass with saturated drift)."""
    rng = np.random.default_rng(9)
    frames = [np.zeros((1024, 1024, 3), np.uint8), np.full((1024, 1024, 3), 255, np.uint8),
              rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)]
    frames = np.ascontiguousarray(np.stack(frames))
    hip_decoder.reset_ccm()
    total, chunks, masks = hip_decoder.decode_batch(frames)
    sym = hip_decoder.tap(D.TAP_SYMBOLS, 3)
    col = hip_decoder.tap(D.TAP_COLORS, 3)
    drift = hip_decoder.tap(D.TAP_DRIFT, 3)
    xy = modeb.cell_positions()
    ccm = pyref.CoCcm()
    for k in range(3):
        r, wchunks, wmask, ccm = pyref.oracle_decode(frames[k], 0, 2, ccm)
        wsym, wcol, wpos = pyref.oracle_stage()
        assert (sym[k] == wsym).all() and (col[k] == wcol).all(), k
        assert (xy + drift[k].astype(np.int32) == wpos).all(), k
        assert masks[k] == wmask and (chunks[k] == wchunks).all(), k
    assert total == 625 * sum(bin(int(m)).count("1") for m in masks)


def test_bad_arguments_are_refused(hip_decoder):
    lib = D.load_library()
    ctx = hip_decoder._ctx
    buf = np.zeros((1024, 1024, 3), np.uint8)
    chunks = np.zeros(7500, np.uint8)
    mask = ctypes.c_uint32(0)
    EINVAL, EDIM = -1, -2
    assert lib.cimbar_hip_decode_frame(ctx, None, 1024, 1024, 3072, 0, 2, chunks.ctypes.data, ctypes.byref(mask)) == EINVAL
    assert lib.cimbar_hip_decode_frame(ctx, buf.ctypes.data, 1024, 1024, 3072, 0, 2, None, ctypes.byref(mask)) == EINVAL
    # too small: the reference's quirk (all-zero RS blocks are valid codewords): full count, full mask, zero chunks
    chunks[:] = 7
    assert lib.cimbar_hip_decode_frame(ctx, buf.ctypes.data, 1024, 1000, 3072, 0, 2, chunks.ctypes.data, ctypes.byref(mask)) == 7500
    assert mask.value == 0xFFF and not chunks.any()
    assert lib.cimbar_hip_decode_frame(ctx, buf.ctypes.data, 1024, 1024, 3000, 0, 2, chunks.ctypes.data, ctypes.byref(mask)) in (EINVAL, EDIM)
    masks = np.zeros(1, np.uint32)
    assert lib.cimbar_hip_decode_batch(ctx, buf.ctypes.data, 0, D.MEM_HOST, 0, 2, chunks.ctypes.data, masks.ctypes.data, D.MEM_HOST, None) == EINVAL
    assert lib.cimbar_hip_decode_batch(ctx, buf.ctypes.data, -3, D.MEM_HOST, 0, 2, chunks.ctypes.data, masks.ctypes.data, D.MEM_HOST, None) == EINVAL
    assert lib.cimbar_hip_decode_batch(ctx, buf.ctypes.data, 1, 7, 0, 2, chunks.ctypes.data, masks.ctypes.data, D.MEM_HOST, None) == EINVAL
    assert len(lib.cimbar_hip_last_error(ctx)) > 0
    # the context is still usable after refusals
    hip_decoder.reset_ccm()
    good, _, m = hip_decoder.decode_frame(buf)
    assert good == 625 * bin(m).count("1")   # (a black frame is all-zero RS codewords: the reference "decodes" some of it too)


def test_unknown_modes_are_mode_b_and_bad_devices_fail_loudly():
    for mode in (5, 12345):   # Config::temp_conf's `case 68: default:` (Config.h:41-43): any value that is no listed mode is mode B, upstream and here
        d = D.HipDecoder(device=0, mode=mode)
        assert d.geo.MODE == 68 and d.bufsize() == 7500
        d.close()
    with pytest.raises(D.CimbarHipError):
        D.HipDecoder(device=4096)
    d = D.HipDecoder(device=0, mode=0)   # 0 = the reference's default config == mode B
    good, _, m = d.decode_frame(np.full((1024, 1024, 3), 255, np.uint8))
    assert good == 625 * bin(m).count("1")


def test_tap_needs_a_decoded_batch():
    d = D.HipDecoder(device=0)
    with pytest.raises(D.CimbarHipError):
        d.tap(D.TAP_SYMBOLS, 1)
