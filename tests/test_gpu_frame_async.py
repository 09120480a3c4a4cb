"""cimbar_hip_decode_frame_async / _wait: one frame per call with several in flight -- the reference's decode loop (cimbar.cpp:124-171, one
Decoder::decode_fountain per image) with frame k+1's host-to-device copy beside frame k's kernels. Every frame's chunks, mask, return value and
the colour-correction carry-over must be what the one-at-a-time loop produces, in the same order."""
import ctypes

import numpy as np
import pytest
import torch

from libcimbar_amd import HipDecoder
from libcimbar_amd import decoder as D
from oracle import pyref
from tests import frames as F

pytestmark = pytest.mark.gpu


def stream_of_frames(synth, n=11, seed=4040):
    """clean, noisy, tinted (their own matrix matters), wiped (no header: the carried matrix matters), shifted (flood path)"""
    payload, fr = F.clean_frames(synth, n, seed=seed)
    out = []
    for k in range(n):
        f = fr[k]
        kind = k % 5
        if kind == 1:
            f = F.add_noise(f, 25, k)
        elif kind == 2:
            f = (f.astype(np.float32) * np.array([0.85, 1.0, 0.7], np.float32)).astype(np.uint8)
        elif kind == 3:
            f = F.blank_region(f, 0, 400, 0, 1024)          # the first symbol chunks (and the header) are gone: colour pass on the carried matrix
        elif kind == 4:
            f = F.shift(f, 1, -2)
        out.append(np.ascontiguousarray(f))
    return payload, out


def one_at_a_time(frames):
    dec = HipDecoder(0)
    total, chunks, masks = dec.decode_batch(np.ascontiguousarray(np.stack(frames)))      # frame order == carry order
    dec.close()
    return chunks, masks


@pytest.mark.parametrize("pinned", [False, True])
def test_frames_in_flight_equal_the_one_at_a_time_loop(synth, pinned):
    payload, frames = stream_of_frames(synth)
    want_chunks, want_masks = one_at_a_time(frames)
    if pinned:
        keep = [torch.from_numpy(f).pin_memory() for f in frames]
        frames = [t.numpy() for t in keep]
    dec = HipDecoder(0)
    depth = dec.pipeline_depth
    tickets, got = [], []
    for k, f in enumerate(frames):
        tickets.append(dec.decode_frame_async(f))
        if len(tickets) - len(got) >= depth:
            got.append(dec.decode_frame_wait(tickets[len(got)]))
    while len(got) < len(tickets):
        got.append(dec.decode_frame_wait(tickets[len(got)]))
    assert tickets == sorted(tickets) and len(set(tickets)) == len(tickets)
    for k, (rc, chunks, mask) in enumerate(got):
        assert mask == want_masks[k], (k, hex(mask), hex(int(want_masks[k])))
        assert (chunks == want_chunks[k]).all(), k
        assert rc == 625 * bin(mask).count("1")
    # and the oracle agrees with the whole stream (carry included)
    ccm = pyref.CoCcm()
    for k, f in enumerate(frames):
        r, wchunks, wmask, ccm = pyref.oracle_decode(f, 0, 2, ccm)
        assert got[k][2] == wmask and (got[k][1] == wchunks).all() and got[k][0] == r, k
    dec.close()


def test_the_synchronous_call_is_the_same_path(synth):
    payload, frames = stream_of_frames(synth, 6, seed=77)
    want_chunks, want_masks = one_at_a_time(frames)
    dec = HipDecoder(0)
    for k, f in enumerate(frames):
        rc, chunks, mask = dec.decode_frame(f)
        assert mask == want_masks[k] and (chunks == want_chunks[k]).all() and rc == 625 * bin(mask).count("1")
    dec.close()


def test_more_frames_started_than_fit_complete_the_oldest_and_keep_its_result(synth):
    """starting frame k + depth delivers frame k into the buffers it was started with; its return value is still there for _wait"""
    payload, frames = stream_of_frames(synth, 8, seed=99)
    want_chunks, want_masks = one_at_a_time(frames)
    dec = HipDecoder(0)
    lib = D.load_library()
    outs = [(np.zeros((12, 625), np.uint8), ctypes.c_uint32(0xDEAD)) for _ in frames]
    tickets = []
    for f, (c, m) in zip(frames, outs):
        t = lib.cimbar_hip_decode_frame_async(dec._ctx, f.ctypes.data, 1024, 1024, 0, 0, 2, c.ctypes.data, ctypes.byref(m))
        assert t >= 0
        tickets.append(t)
    depth = dec.pipeline_depth
    for k in range(len(frames) - depth):              # delivered by the calls that needed their slot
        assert outs[k][1].value == want_masks[k] and (outs[k][0] == want_chunks[k]).all(), k
    for k, t in enumerate(tickets):
        rc = lib.cimbar_hip_decode_frame_wait(dec._ctx, t)
        assert rc == 625 * bin(int(want_masks[k])).count("1"), (k, rc)
        assert outs[k][1].value == want_masks[k] and (outs[k][0] == want_chunks[k]).all(), k
    assert lib.cimbar_hip_decode_frame_wait(dec._ctx, tickets[-1] + 5) < 0          # never issued
    assert lib.cimbar_hip_decode_frame_wait(dec._ctx, -1) < 0
    dec.close()


def test_other_image_sizes_and_strides_inside_a_stream(synth):
    """a padded image (CimbReader's _gridPadding case), a too-small one and a strided one between ordinary frames: same results and same carry
    as the synchronous calls in the same order"""
    payload, frames = stream_of_frames(synth, 5, seed=123)
    big = np.zeros((1040, 1060, 3), np.uint8)
    big[8:1032, 8:1032] = frames[1]
    small = np.zeros((500, 500, 3), np.uint8)
    strided = np.full((1024, 1024 * 3 + 96), 0x5A, np.uint8)
    strided[:, :3072] = frames[3].reshape(1024, -1)
    seq = [frames[0], big, small, frames[2], strided, frames[4]]

    def run(dec, use_async):
        lib = D.load_library()
        res, tickets, outs = [], [], []
        for img in seq:
            c, m = np.zeros((12, 625), np.uint8), ctypes.c_uint32(0)
            if img.ndim == 2:
                args = (img.ctypes.data, 1024, 1024, img.strides[0])
            else:
                args = (img.ctypes.data, img.shape[1], img.shape[0], img.strides[0])
            if use_async:
                t = lib.cimbar_hip_decode_frame_async(dec._ctx, *args, 0, 2, c.ctypes.data, ctypes.byref(m))
                assert t >= 0
                tickets.append(t)
                outs.append((c, m))
            else:
                rc = lib.cimbar_hip_decode_frame(dec._ctx, *args, 0, 2, c.ctypes.data, ctypes.byref(m))
                res.append((rc, c.copy(), m.value))
        for t, (c, m) in zip(tickets, outs):
            rc = lib.cimbar_hip_decode_frame_wait(dec._ctx, t)
            res.append((rc, c.copy(), m.value))
        return res
    a, b = HipDecoder(0), HipDecoder(0)
    ra, rb = run(a, True), run(b, False)
    for k, (x, y) in enumerate(zip(ra, rb)):
        assert x[0] == y[0] and x[2] == y[2] and (x[1] == y[1]).all(), k
    assert ra[2][0] == 7500 and ra[2][2] == 0xFFF and not ra[2][1].any()          # the too-small image: the reference's all-zero chunks
    a.close()
    b.close()


def test_batches_and_frames_interleave_on_one_context(synth):
    """a pipelined / ordinary batch between frames in flight: every entry point waits for what it must"""
    payload, frames = stream_of_frames(synth, 6, seed=321)
    want_chunks, want_masks = one_at_a_time(frames)
    dec = HipDecoder(0)
    t0 = dec.decode_frame_async(frames[0])
    t1 = dec.decode_frame_async(frames[1])
    total, chunks, masks = dec.decode_batch(np.ascontiguousarray(np.stack(frames[2:4])))        # carries on from frame 1's matrix
    t4 = dec.decode_frame_async(frames[4])
    t5 = dec.decode_frame_async(frames[5])
    r = [dec.decode_frame_wait(t) for t in (t0, t1)] + [(None, chunks[0], int(masks[0])), (None, chunks[1], int(masks[1]))] + \
        [dec.decode_frame_wait(t) for t in (t4, t5)]
    for k in range(6):
        assert r[k][2] == want_masks[k] and (r[k][1] == want_chunks[k]).all(), k
    dec.close()


@pytest.mark.parametrize("mode", [67, 66, 4, 8])
def test_frames_in_flight_in_every_mode(mode):
    """the other geometries (1024x720, 736x637) and the legacy single-stream modes through the same entry points: equal to the batch call"""
    from libcimbar_amd import framegen
    synth_m = framegen.FrameSynth("cpu", mode)
    payload = framegen.synth_payload(7, seed=900 + mode, mode=mode)
    frames = synth_m.frames_from_payload(payload).numpy()
    frames = [np.ascontiguousarray(F.add_noise(f, 12, k) if k % 2 else f) for k, f in enumerate(frames)]
    ref = HipDecoder(0, mode)
    total, want_chunks, want_masks = ref.decode_batch(np.ascontiguousarray(np.stack(frames)))
    ref.close()
    dec = HipDecoder(0, mode)
    tickets = [dec.decode_frame_async(f) for f in frames]          # more than fit: the oldest are completed by the calls that need their slot
    for k, t in enumerate(tickets):
        rc, chunks, mask = dec.decode_frame_wait(t)
        assert mask == want_masks[k] and (chunks == want_chunks[k]).all(), (mode, k)
        assert rc == dec.geo.CHUNK * bin(mask).count("1")
    assert (np.stack([want_chunks[k].reshape(-1) for k in range(0, 7, 2)]) == payload.numpy()[0::2]).all()      # the clean ones decode to what was encoded
    dec.close()


def test_destroying_a_context_with_frames_in_flight(synth):
    """frames started and never waited for: destroy waits for the device before it frees what the kernels write into; a fresh context works afterwards"""
    payload, frames = stream_of_frames(synth, 4, seed=11)
    want_chunks, want_masks = one_at_a_time(frames)
    dec = HipDecoder(0)
    for f in frames[:3]:
        dec.decode_frame_async(f)
    dec.close()
    dec = HipDecoder(0)
    rc, chunks, mask = dec.decode_frame(frames[0])
    assert mask == want_masks[0] and (chunks == want_chunks[0]).all()
    dec.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_interleaving_of_every_entry_point_keeps_frame_order_semantics(synth, seed):
    """One context, a seeded random sequence of: frames started / waited (in and out of ticket order), synchronous frames, ordinary batches, pipelined
    batches from device memory. Whatever the interleaving, frame k's chunks, mask and colour-correction carry must be those of decoding the same frames
    one after the other in submission order (what a single reference thread produces)."""
    rng = np.random.default_rng(1000 + seed)
    payload, frames = stream_of_frames(synth, 24, seed=700 + seed)
    want_chunks, want_masks = one_at_a_time(frames)
    dev = torch.device("cuda", 0)
    dec = HipDecoder(0)
    st = torch.cuda.current_stream(dev)
    got = {}
    pending = []          # (ticket, frame index)
    device_batches = []   # (first index, count, chunks tensor, masks tensor, frames tensor)
    k = 0
    while k < len(frames):
        op = int(rng.integers(0, 5))
        if op == 0:                                   # start one frame
            pending.append((dec.decode_frame_async(frames[k]), k))
            k += 1
        elif op == 1 and pending:                     # wait for a random pending frame (not necessarily the oldest)
            t, idx = pending.pop(int(rng.integers(0, len(pending))))
            rc, c, m = dec.decode_frame_wait(t)
            got[idx] = (c.copy(), m)
        elif op == 2:                                 # a synchronous frame
            rc, c, m = dec.decode_frame(frames[k])
            got[k] = (c.copy(), m)
            k += 1
        elif op == 3:                                 # an ordinary batch of 1-3 frames from host memory
            n = min(int(rng.integers(1, 4)), len(frames) - k)
            total, c, m = dec.decode_batch(np.ascontiguousarray(np.stack(frames[k:k + n])))
            for j in range(n):
                got[k + j] = (c[j].copy(), int(m[j]))
            k += n
        elif op == 4:                                 # a pipelined batch of 1-2 frames from device memory
            n = min(int(rng.integers(1, 3)), len(frames) - k)
            fr = torch.from_numpy(np.ascontiguousarray(np.stack(frames[k:k + n]))).to(dev)
            c = torch.zeros((n, 7500), dtype=torch.uint8, device=dev)
            m = torch.zeros(n, dtype=torch.int32, device=dev)
            dec.decode_batch_pipelined(fr.data_ptr(), n, c.data_ptr(), m.data_ptr(), False, 2, st.cuda_stream)
            device_batches.append((k, n, c, m, fr))
            k += n
        # never more tickets outstanding than the library keeps results for
        while len(pending) > 8:
            t, idx = pending.pop(0)
            rc, c, m = dec.decode_frame_wait(t)
            got[idx] = (c.copy(), m)
    for t, idx in pending:
        rc, c, m = dec.decode_frame_wait(t)
        got[idx] = (c.copy(), m)
    dec.pipeline_wait(st.cuda_stream)
    torch.cuda.synchronize(dev)
    for first, n, c, m, _fr in device_batches:
        for j in range(n):
            got[first + j] = (c[j].cpu().numpy().reshape(12, 625), int(m[j].item()) & 0xFFFFFFFF)
    assert sorted(got) == list(range(len(frames)))
    for idx in range(len(frames)):
        c, m = got[idx]
        assert m == want_masks[idx], (seed, idx, hex(m), hex(int(want_masks[idx])))
        assert (np.asarray(c).reshape(12, 625) == want_chunks[idx]).all(), (seed, idx)
    dec.close()


def test_a_pageable_image_may_be_overwritten_as_soon_as_the_call_returns(synth):
    """include/cimbar_hip.h: a pageable `rgb` may be reused or freed at once -- also when it is a strided view (a cv::Mat ROI), which would otherwise go
    through hipMemcpy2DAsync with the DMA possibly still pending at return. Dense and strided images, scribbled over right after the call."""
    payload, frames = stream_of_frames(synth, 6, seed=99)
    want_chunks, want_masks = one_at_a_time(frames)
    dec = HipDecoder(0)
    for strided in (False, True):
        dec.reset_ccm()
        tickets, outs = [], []
        for k, f in enumerate(frames):
            if strided:
                wide = np.zeros((1024, 1024 * 3 + 192), np.uint8)          # the image is a window of a wider pageable buffer
                wide[:, 96:96 + 3072] = f.reshape(1024, 3072)
                view = wide[:, 96:96 + 3072]
                ptr, stride, owner = view.ctypes.data, wide.strides[0], wide
            else:
                owner = f.copy()
                ptr, stride = owner.ctypes.data, 0
            chunks = np.zeros((12, 625), np.uint8)
            mask = ctypes.c_uint32(0)
            t = dec._lib.cimbar_hip_decode_frame_async(dec._ctx, ptr, 1024, 1024, stride, 0, 2, chunks.ctypes.data, ctypes.byref(mask))
            assert t >= 0
            owner[:] = 0xA5                                                 # the caller's buffer is reused at once
            tickets.append(int(t))
            outs.append((chunks, mask))
        for k, t in enumerate(tickets):
            rc = dec._lib.cimbar_hip_decode_frame_wait(dec._ctx, t)
            chunks, mask = outs[k]
            assert mask.value == want_masks[k] and (chunks == want_chunks[k]).all() and rc == 625 * bin(mask.value).count("1"), (strided, k)
    dec.close()
