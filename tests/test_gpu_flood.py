"""GPU: the exact flood pass (k_flood) in its own right -- frames that leave the fast path, compared with the oracle cell by
cell, through the normal library and through the spill-test build (libcimbar_hip_spilltest.so: the same source compiled with
CIMBAR_HEAP_LDS=1024, so that most heap operations cross the LDS / global-scratch boundary)."""
import os

import numpy as np
import pytest

from libcimbar_amd import build as hipbuild
from libcimbar_amd import decoder as D
from libcimbar_amd import modeb
from oracle import pyref
from tests import frames as F

pytestmark = pytest.mark.gpu


def exact_decoder(lib_path=None):
    """a decoder whose flagged frames ALL take the exact replay (k_flood): the batch-parallel pass in front of it is switched off"""
    old = os.environ.get("CIMBAR_HIP_FLOOD_WAVE")
    os.environ["CIMBAR_HIP_FLOOD_WAVE"] = "0"
    try:
        return D.HipDecoder(0, lib_path=lib_path)
    finally:
        if old is None:
            del os.environ["CIMBAR_HIP_FLOOD_WAVE"]
        else:
            os.environ["CIMBAR_HIP_FLOOD_WAVE"] = old


def flood_frames(synth):
    payload, frames = F.clean_frames(synth, 4, seed=91)
    rng = np.random.default_rng(3)
    return [
        F.shift(frames[0], 2, 1),                                       # rigid shift: every cell drifts, all priorities equal
        F.add_noise(F.shift(frames[1], -3, 2), 40, 7),                  # drift + noise: mixed priorities, stale heap entries
        F.rescale(frames[2], 6),                                          # drift grows across the frame
        rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8),          # pure noise: 9-window mode nearly everywhere
    ], ["shift", "shift+noise", "rescale", "noise"]


def check(dec, frames, names):
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    dec.reset_ccm()
    total, chunks, masks = dec.decode_batch(frames)
    assert dec.tap(D.TAP_FLOOD, n).all(), "these frames are meant to leave the fast path"
    sym, col, drift = dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_COLORS, n), dec.tap(D.TAP_DRIFT, n)
    xy = modeb.cell_positions()
    ccm = pyref.CoCcm()
    for k in range(n):
        r, wchunks, wmask, ccm = pyref.oracle_decode(frames[k], 0, 2, ccm)
        wsym, wcol, wpos = pyref.oracle_stage()
        assert (sym[k] == wsym).all(), f"{names[k]}: {(sym[k] != wsym).sum()} symbols differ"
        assert (xy + drift[k].astype(np.int32) == wpos).all(), f"{names[k]}: drifted positions differ"
        assert (col[k] == wcol).all(), f"{names[k]}: colours differ"
        assert masks[k] == wmask and (chunks[k] == wchunks).all(), names[k]


def test_flood_pass_matches_oracle(synth):
    frames, names = flood_frames(synth)
    dec = exact_decoder()
    check(dec, frames, names)
    assert (dec.tap(D.TAP_FLOOD_PATH, len(frames)) == 1).all()


def test_flood_through_the_default_path_matches_oracle(hip_decoder, synth):
    """the same frames through the library as shipped: the batch-parallel flood (k_flood_wave) takes what it can certify, the exact
    replay the rest -- results must not depend on which one ran"""
    frames, names = flood_frames(synth)
    check(hip_decoder, frames, names)
    path = hip_decoder.tap(D.TAP_FLOOD_PATH, len(frames))
    assert path[0] == 2, "a clean, rigidly shifted frame is the textbook case for the batch-parallel flood"
    assert path[3] == 1, "pure noise cannot be certified"


def wave_frames(synth):
    payload, fr = F.clean_frames(synth, 6, seed=191)
    return [
        F.shift(fr[0], 2, 1), F.shift(fr[1], 1, 0), F.shift(fr[2], -1, -1), F.shift(fr[3], 0, 3),
        F.blank_region(F.shift(fr[4], 1, 1), 300, 420, 0, 1024),              # a wiped band: the wave has to go around / through it
        F.blank_region(fr[5], 60, 500, 60, 700),                              # unshifted, a wiped block: flagged by its garbage cells
        F.blank_region(F.shift(fr[0], -2, 1), 0, 1024, 0, 560, value=255),    # half the frame gone
        F.add_noise(F.shift(fr[1], 1, 2), 12, 5),                             # light noise on a shift
    ], ["shift+2+1", "shift+1+0", "shift-1-1", "shift0+3", "shift+band", "block", "shift+half", "shift+noise12"]


def test_batch_parallel_flood_matches_oracle_and_certifies_clean_shifts(hip_decoder, synth):
    frames, names = wave_frames(synth)
    check(hip_decoder, frames, names)
    path = hip_decoder.tap(D.TAP_FLOOD_PATH, len(frames))
    assert (path[:4] == 2).all(), f"clean rigid shifts must certify: {path}"
    assert (path != 0).all()


def test_batch_parallel_flood_and_exact_replay_agree(hip_decoder, synth):
    """every frame of a larger, mixed batch: default path vs exact replay, cell by cell (no oracle: this is GPU vs GPU at a batch size the
    oracle would take minutes for)"""
    payload, fr = F.clean_frames(synth, 8, seed=404)
    g = np.random.default_rng(8)
    frames = []
    for k in range(48):
        f = F.shift(fr[k % 8], int(g.integers(-3, 4)), int(g.integers(-3, 4)))
        if k % 3 == 1:
            y0, x0 = int(g.integers(0, 800)), int(g.integers(0, 800))
            f = F.blank_region(f, y0, y0 + int(g.integers(20, 300)), x0, x0 + int(g.integers(20, 300)), value=int(g.integers(0, 256)))
        if k % 3 == 2:
            f = F.add_noise(f, int(g.integers(4, 60)), k)
        frames.append(f)
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    ex = exact_decoder()
    outs = []
    for dec in (hip_decoder, ex):
        dec.reset_ccm()
        total, chunks, masks = dec.decode_batch(frames)
        outs.append((chunks.copy(), masks.copy(), dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_DRIFT, n), dec.tap(D.TAP_COLORS, n), dec.tap(D.TAP_FLOOD_PATH, n)))
    a, b = outs
    assert (b[5] <= 1).all()
    flagged = a[5] != 0
    assert ((a[5] != 0) == (b[5] != 0)).all()
    for k in range(n):
        assert (a[2][k] == b[2][k]).all(), f"frame {k} (path {a[5][k]}): symbols differ"
        if flagged[k]:
            assert (a[3][k] == b[3][k]).all(), f"frame {k} (path {a[5][k]}): drift differs"
        assert (a[4][k] == b[4][k]).all() and a[1][k] == b[1][k] and (a[0][k] == b[0][k]).all(), k
    assert (a[5] == 2).sum() >= 8, f"too few frames certified: {a[5]}"


def test_flood_pass_with_the_heap_spilling_to_global_memory(synth):
    assert os.path.exists(hipbuild.OUT_SPILLTEST), "build it with `python -m libcimbar_amd.build` (or __graft_entry__.build())"
    dec = exact_decoder(lib_path=hipbuild.OUT_SPILLTEST)
    frames, names = flood_frames(synth)
    check(dec, frames, names)


def test_single_frame_batches_and_repeat_calls(hip_decoder, synth):
    """the flood kernel is persistent over a strided frame list: batch sizes around its grid and repeated calls agree"""
    payload, frames = F.clean_frames(synth, 3, seed=17)
    shifted = np.ascontiguousarray(np.stack([F.shift(f, 1, -2) for f in frames]))
    hip_decoder.reset_ccm()
    _, c3, m3 = hip_decoder.decode_batch(shifted, color_correction=0)
    for k in range(3):
        _, c1, m1 = hip_decoder.decode_batch(shifted[k:k + 1], color_correction=0)
        assert m1[0] == m3[k] and (c1[0] == c3[k]).all()
    assert (m3 == 0xFFF).all() and (c3.reshape(3, -1) == payload).all()


def test_split_batch_floods_with_spilling_heaps(synth):
    """a batch large enough to run as two half-batches on two streams: both halves' flood kernels spill at the same time"""
    import torch
    from libcimbar_amd import framegen
    dev = torch.device("cuda", 0)
    dec = exact_decoder(lib_path=hipbuild.OUT_SPILLTEST)
    n = 160
    payload = framegen.synth_payload(n, seed=11, device=dev)
    frames = torch.roll(framegen.FrameSynth(dev).frames_from_payload(payload), shifts=(1, 2), dims=(1, 2)).contiguous()
    chunks = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    dec.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert int(dec.tap(D.TAP_FLOOD, n).sum()) == n
    assert bool((masks == 0xFFF).all()) and bool((chunks == payload).all())


@pytest.mark.parametrize("mode", [68, 67, 66])
def test_certified_batch_flood_fuzz_against_exact_replay(mode):
    """256 randomly distorted frames per mode (shifts, wipes, noise, tears between two shifts, random 9x9 patches): wherever the batch-parallel
    flood certifies a frame, its symbols and drift equal the exact replay's -- the soundness claim of k_flood_wave, on the kernel itself"""
    import os
    import torch
    from libcimbar_amd import HipDecoder, framegen
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    synth = framegen.FrameSynth("cpu", mode)
    payload, fr = F.clean_frames(synth, 8, seed=500 + mode)
    h, w = fr.shape[1:3]
    g = np.random.default_rng(mode)
    frames = []
    for k in range(256):
        f = F.shift(fr[k % 8], int(g.integers(-3, 4)), int(g.integers(-3, 4)))
        kind = k % 6
        if kind == 1:
            y0, x0 = int(g.integers(0, h - 100)), int(g.integers(0, w - 100))
            f = F.blank_region(f, y0, y0 + int(g.integers(20, 300)), x0, x0 + int(g.integers(20, 300)), value=int(g.integers(0, 256)))
        elif kind == 2:
            f = F.add_noise(f, int(g.integers(3, 40)), k)
        elif kind == 3:
            cut = int(g.integers(100, h - 100))
            f = np.concatenate([F.shift(fr[k % 8], 1, 0)[:cut], F.shift(fr[k % 8], 0, 1)[cut:]], 0)     # a tear: two different shifts
        elif kind == 4:
            f = f.copy()
            for _ in range(int(g.integers(1, 40))):
                y, x = int(g.integers(8, h - 20)), int(g.integers(8, w - 20))
                f[y:y + 9, x:x + 9] = g.integers(0, 256, (9, 9, 3))
        elif kind == 5:
            f = F.shift(fr[k % 8], int(g.integers(-7, 8)), int(g.integers(-7, 8)))                       # up to the drift limit
        frames.append(f)
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    outs = []
    for wave in ("1", "0"):
        os.environ["CIMBAR_HIP_FLOOD_WAVE"] = wave
        try:
            dec = HipDecoder(0, mode)
        finally:
            os.environ.pop("CIMBAR_HIP_FLOOD_WAVE", None)
        total, chunks, masks = dec.decode_batch(frames)
        outs.append((chunks.copy(), masks.copy(), dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_DRIFT, n), dec.tap(D.TAP_FLOOD_PATH, n)))
        dec.close()
    a, b = outs
    assert (b[4] <= 1).all() and ((a[4] != 0) == (b[4] != 0)).all()
    for k in range(n):
        assert (a[2][k] == b[2][k]).all(), f"mode {mode} frame {k} (path {a[4][k]}): symbols differ"
        if a[4][k]:
            assert (a[3][k] == b[3][k]).all(), f"mode {mode} frame {k} (path {a[4][k]}): drift differs"
        assert a[1][k] == b[1][k] and (a[0][k] == b[0][k]).all(), k
    assert (a[4] == 2).sum() >= 40, f"too few frames certified: {np.bincount(a[4])}"
