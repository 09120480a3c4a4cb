"""Images LARGER than the mode's frame (CimbReader.cpp:112-117: the grid sits _gridPadding = min(cols - image_size_x, rows - image_size_y) / 2 pixels
in, and the threshold pass runs over the whole image) and images SMALLER than it (CimbReader.cpp:119: nothing is read, the Reed-Solomon pass
delivers twelve chunks of zeros). CPU half: the oracle against the reference build. GPU half: cimbar_hip_decode_frame against the oracle."""
import numpy as np
import pytest

from libcimbar_amd import framegen
from oracle import pyref
from tests import frames as F

SIZES = {68: (1024, 1024), 67: (1024, 720), 66: (736, 637), 4: (1024, 1024), 8: (1024, 1024)}
EXTRA = [(16, 16), (10, 30), (7, 3), (64, 0), (1, 1), (0, 9), (301, 57)]


def padded_cases(mode, seed=3):
    """(name, image) list: clean and distorted frames set into random-pixel borders of several extra widths / heights"""
    w, h = SIZES[mode]
    synth = framegen.FrameSynth("cpu", mode)
    _, fr = F.clean_frames(synth, 3, seed=seed)
    variants = [("clean", fr[0]), ("noisy-shift", F.add_noise(F.shift(fr[1], 2, -1), 40, 1)), ("rescale", F.rescale(fr[2], 6))]
    rng = np.random.default_rng(seed)
    out = []
    for (ew, eh) in EXTRA:
        for nm, frame in variants:
            big = rng.integers(0, 256, (h + eh, w + ew, 3), dtype=np.uint8)
            pad = min(ew, eh) // 2
            big[pad:pad + h, pad:pad + w] = frame
            if nm == "clean":          # bright pixels right against the grid: the box mean of the outermost cells sees them
                big[:pad] = 255
                big[:, :pad] = 255
            out.append((f"{nm}+{ew}x{eh}", big))
    return out


@pytest.mark.parametrize("mode", [68, 67, 66, 4, 8])
def test_oracle_matches_reference_on_padded_and_small_images(ref, mode):
    w, h = SIZES[mode]
    with pyref.ref_mode(mode):
        for pre in (0, 1):
            ccm = pyref.CoCcm()
            for k, (nm, big) in enumerate(padded_cases(mode)):
                r, ch, m = pyref.ref_decode(big, pre, 2, reset_ccm=(k == 0), mode=mode)
                r2, ch2, m2, ccm = pyref.oracle_decode(big, pre, 2, ccm, mode=mode)
                assert (r, m) == (r2, m2) and (ch == ch2).all(), (nm, pre)
        for shape in ((h - 1, w, 3), (h, w - 8, 3), (100, 100, 3)):
            small = np.random.default_rng(1).integers(0, 256, shape, dtype=np.uint8)
            r, ch, m = pyref.ref_decode(small, 0, 2, mode=mode)
            r2, ch2, m2, _ = pyref.oracle_decode(small, 0, 2, None, mode=mode)
            assert (r, m) == (r2, m2) == (ch.shape[0] * ch.shape[1], (1 << ch.shape[0]) - 1) and not ch.any() and not ch2.any()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [68, 67, 66, 4, 8])
def test_gpu_padded_and_small_images_match_oracle(mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from libcimbar_amd import HipDecoder
    dec = HipDecoder(0, mode)
    w, h = SIZES[mode]
    for pre in (0, 1):
        ccm = pyref.CoCcm()
        dec.reset_ccm()
        decoded = 0
        for nm, big in padded_cases(mode):
            r, ch, m = dec.decode_frame(big, should_preprocess=pre, color_correction=2)
            r2, ch2, m2, ccm = pyref.oracle_decode(big, pre, 2, ccm, mode=mode)
            assert (r, m) == (r2, m2) and (ch == ch2).all(), (nm, pre)
            decoded += m == dec.geo.FULL_MASK
        assert decoded >= len(EXTRA)
    # row stride larger than the row, and a too-small image
    nm, big = padded_cases(mode)[3]
    wide = np.zeros((big.shape[0], big.shape[1] + 5, 3), np.uint8)
    wide[:, :big.shape[1]] = big
    view = wide[:, :big.shape[1]]                 # same pixels, stride = (w + 5) * 3
    lib, ctx = dec._lib, dec._ctx
    import ctypes
    chunks = np.zeros((dec.geo.CHUNKS_PER_FRAME, dec.geo.CHUNK), np.uint8)
    mask = ctypes.c_uint32(0)
    dec.reset_ccm()
    r = lib.cimbar_hip_decode_frame(ctx, view.ctypes.data, view.shape[1], view.shape[0], view.strides[0], 0, 2, chunks.ctypes.data, ctypes.byref(mask))
    r2, ch2, m2, _ = pyref.oracle_decode(big, 0, 2, None, mode=mode)
    assert (r, mask.value) == (r2, m2) and (chunks == ch2).all()
    r, ch, m = dec.decode_frame(np.full((h - 1, w, 3), 9, np.uint8))
    assert (r, m) == (dec.geo.FRAME_BYTES, dec.geo.FULL_MASK) and not ch.any()
    dec.close()
