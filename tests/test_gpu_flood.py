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


def test_flood_pass_matches_oracle(hip_decoder, synth):
    frames, names = flood_frames(synth)
    check(hip_decoder, frames, names)


def test_flood_pass_with_the_heap_spilling_to_global_memory(synth):
    assert os.path.exists(hipbuild.OUT_SPILLTEST), "build it with `python -m libcimbar_amd.build` (or __graft_entry__.build())"
    dec = D.HipDecoder(0, lib_path=hipbuild.OUT_SPILLTEST)
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
    dec = D.HipDecoder(0, lib_path=hipbuild.OUT_SPILLTEST)
    n = 160
    payload = framegen.synth_payload(n, seed=11, device=dev)
    frames = torch.roll(framegen.FrameSynth(dev).frames_from_payload(payload), shifts=(1, 2), dims=(1, 2)).contiguous()
    chunks = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    dec.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert int(dec.tap(D.TAP_FLOOD, n).sum()) == n
    assert bool((masks == 0xFFF).all()) and bool((chunks == payload).all())
