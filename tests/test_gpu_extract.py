"""GPU: the stage in front of the decoder -- Scanner's image preparation and Deskewer's warp on synthetic camera
captures, against the oracle's restatement (bit-exact), and the whole chain capture -> GPU preprocess -> [reference Scanner on the
host] -> GPU deskew -> GPU decode against the reference's Extractor + Decoder. Parity with a real OpenCV is unpinned."""
import ctypes
import os

import numpy as np
import pytest
import torch

from libcimbar_amd import decoder as D
from libcimbar_amd import modeb
from oracle import pyref
from oracle.pyref import P
from tests import frames as F
from tests.test_oracle_vs_ref import CAMERA_CASES

pytestmark = pytest.mark.gpu


def captures(synth, size=(1920, 1080)):
    payload, frames = F.clean_frames(synth, len(CAMERA_CASES), seed=90)
    cams = [F.camera_frame(frames[k], width=size[0], height=size[1], quad=q, background=bg, blur=bl) for k, (bg, q, bl) in enumerate(CAMERA_CASES)]
    return payload, np.ascontiguousarray(np.stack(cams))


def test_scan_preprocess_matches_oracle(hip_decoder, synth, oracle):
    _, cams = captures(synth)
    got, thr = hip_decoder.scan_preprocess(cams)
    for k in range(len(cams)):
        want = np.zeros(cams[k].shape[:2], np.uint8)
        t = oracle.co_scan_preprocess(P(cams[k]), cams.shape[2], cams.shape[1], P(want))
        assert thr[k] == t
        assert (got[k] == want).all(), f"capture {k}: {(got[k] != want).sum()} pixels differ"


def test_scan_preprocess_5x5_blur_and_odd_sizes(hip_decoder, oracle):
    rng = np.random.default_rng(4)
    for (w, h) in ((2600, 1700), (1283, 977), (64, 40)):          # 5x5 kernel above 1500 px; sizes that are no multiple of the tile
        img = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
        img[0, : h // 2] //= 3
        got, thr = hip_decoder.scan_preprocess(img)
        want = np.zeros((h, w), np.uint8)
        assert thr[0] == oracle.co_scan_preprocess(P(img[0]), w, h, P(want))
        assert (got[0] == want).all(), (w, h)


def test_deskew_matches_oracle_and_reference_chain(hip_decoder, synth, oracle, ref):
    payload, cams = captures(synth)
    n, h, w = cams.shape[:3]
    corners = np.zeros((n, 8), np.float32)
    for k in range(n):          # the anchor search is the reference's host code, fed with the capture as the reference feeds it
        c = (ctypes.c_float * 8)()
        assert ref.ref_scan_corners(P(cams[k]), w, h, c) == 4
        corners[k] = list(c)
    frames = hip_decoder.deskew_batch(cams, corners)
    for k in range(n):
        want = np.zeros((1024, 1024, 3), np.uint8)
        oracle.co_deskew(P(cams[k]), w, h, corners[k].ctypes.data_as(ctypes.POINTER(ctypes.c_float)), P(want))
        assert (frames[k] == want).all(), f"capture {k}: {(frames[k] != want).sum()} bytes differ"
    # ... and on through the decoder: same chunks as Extractor::extract + Decoder::decode_fountain of the reference build
    full = 0
    for k in range(n):
        ext = np.zeros((1024, 1024, 3), np.uint8)
        rc = ref.ref_extract(P(cams[k]), w, h, P(ext))
        assert rc in (1, 2) and (ext == frames[k]).all()
        r, chunks, mask = pyref.ref_decode(ext, rc == 2, 2, 1)
        hip_decoder.reset_ccm()
        good, got, gmask = hip_decoder.decode_frame(frames[k], rc == 2, 2)
        assert (good, gmask) == (r, mask) and (got == chunks).all()
        full += int(good == 7500 and (got.reshape(-1) == payload[k]).all())
    assert full >= 1      # (whether a given capture survives is the reference's business; agreeing with it is ours)


def test_deskew_to_decode_without_leaving_the_device(hip_decoder, synth, ref):
    payload, cams = captures(synth)
    n, h, w = cams.shape[:3]
    corners = np.zeros((n, 8), np.float32)
    for k in range(n):
        c = (ctypes.c_float * 8)()
        assert ref.ref_scan_corners(P(cams[k]), w, h, c) == 4
        corners[k] = list(c)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    d_cams = torch.from_numpy(cams).to(dev)
    d_frames = torch.empty((n, 1024, 1024, 3), dtype=torch.uint8, device=dev)
    chunks = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    hip_decoder.reset_ccm()
    hip_decoder.deskew_batch_device(d_cams.data_ptr(), w, h, n, corners, d_frames.data_ptr(), st)
    hip_decoder.decode_batch_device(d_frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), True, 2, st)   # upscaled captures: sharpen
    torch.cuda.synchronize()
    hip_decoder.reset_ccm()
    for k in range(n):          # the same frames through the host entry points
        fr = hip_decoder.deskew_batch(cams[k:k + 1], corners[k:k + 1])
        assert (fr[0] == d_frames[k].cpu().numpy()).all()
    want_total, want_chunks, want_masks = hip_decoder.decode_batch(d_frames.cpu().numpy(), True, 2)
    assert (want_masks.astype(np.int64) == masks.cpu().numpy().astype(np.int64)).all()
    assert (want_chunks.reshape(n, -1) == chunks.cpu().numpy()).all()


def test_extract_stage_bad_arguments(hip_decoder):
    lib = D.load_library()
    buf = np.zeros((40, 64, 3), np.uint8)
    out = np.zeros((40, 64), np.uint8)
    assert lib.cimbar_hip_scan_preprocess(hip_decoder._ctx, None, 64, 40, 1, D.MEM_HOST, out.ctypes.data, None, D.MEM_HOST, None) == -1
    assert lib.cimbar_hip_scan_preprocess(hip_decoder._ctx, buf.ctypes.data, 64, 40, 0, D.MEM_HOST, out.ctypes.data, None, D.MEM_HOST, None) == -1
    assert lib.cimbar_hip_deskew_batch(hip_decoder._ctx, buf.ctypes.data, 64, 40, 1, D.MEM_HOST, None, out.ctypes.data, D.MEM_HOST, None) == -1


@pytest.mark.parametrize("size,quad", [((1280, 720), ((300, 10), (990, 20), (290, 700), (1000, 690))),
                                        ((3840, 2160), ((900, 60), (2950, 100), (880, 2090), (2980, 2050))),
                                        ((4000, 2600), ((800, 70), (3300, 120), (780, 2500), (3340, 2460))),      # short side >= 2500 px: 9x9 blur
                                        ((5600, 4600), ((600, 90), (4900, 160), (560, 4480), (4960, 4400)))])     # short side >= 4500 px: 17x17 blur
def test_other_capture_sizes_match_oracle(hip_decoder, synth, oracle, size, quad):
    """720p (3x3 blur, the warp upscales) and 2160p (5x5 blur, the warp shrinks): both passes against the oracle, corners from the quad"""
    w, h = size
    _, frames = F.clean_frames(synth, 1, seed=77)
    cam = np.ascontiguousarray(F.camera_frame(frames[0], width=w, height=h, quad=quad, background=0))
    got, thr = hip_decoder.scan_preprocess(cam[None])
    want = np.zeros((h, w), np.uint8)
    assert thr[0] == oracle.co_scan_preprocess(P(cam), w, h, P(want))
    assert (got[0] == want).all()
    corners = np.array([float(v) for p in quad for v in p], np.float32)
    desk = hip_decoder.deskew_batch(cam[None], corners[None])
    wantd = np.zeros((1024, 1024, 3), np.uint8)
    oracle.co_deskew(P(cam), w, h, corners.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), P(wantd))
    assert (desk[0] == wantd).all(), f"{(desk[0] != wantd).sum()} bytes differ"
    # corners that push part of the frame outside the capture: out-of-image taps read 0
    far = corners.copy(); far[0] -= 400; far[1] -= 300
    desk = hip_decoder.deskew_batch(cam[None], far[None])
    oracle.co_deskew(P(cam), w, h, far.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), P(wantd))
    assert (desk[0] == wantd).all()


@pytest.fixture(scope="module")
def tiny_list_decoder():
    """the test build of the library (libcimbar_hip_spilltest.so, -DCIMBAR_SCAN_TINY_LISTS): the anchor search keeps ONE hit per scan row, so
    any capture with two anchors on a row overflows the fast kernels' lists and goes through the serial slow path (k_scan_serial)"""
    from libcimbar_amd import build as hipbuild
    assert os.path.exists(hipbuild.OUT_SPILLTEST), "build it with `python -m libcimbar_amd.build` (or __graft_entry__.build())"
    d = D.HipDecoder(0, lib_path=hipbuild.OUT_SPILLTEST)
    yield d
    d.close()


@pytest.mark.parametrize("which", ["fast", "serial-slow-path"])
def test_extract_batch_matches_oracle_on_captures_that_work_and_that_fail(hip_decoder, tiny_list_decoder, synth, oracle, which):
    """Extractor::extract entirely on the device (anchor search included): status, corners and deskewed frames == the oracle's co_extract
    (itself pinned to the reference's Scanner / Extractor in tests/test_oracle_vs_ref.py), per capture size -- through the fast kernels, and
    with their lists shrunk to one entry so that the search overflows and k_scan_serial (Scanner::scan on one lane, lists in global memory)
    has to produce the same answers"""
    from tests.test_oracle_vs_ref import scan_cases
    if which != "fast":
        hip_decoder = tiny_list_decoder
    cams = scan_cases(synth)
    by_size = {}
    for k, cam in enumerate(cams):
        by_size.setdefault(cam.shape, []).append(k)
    ok = 0
    for shape, idx in by_size.items():
        batch = np.ascontiguousarray(np.stack([cams[k] for k in idx]))
        status, corners, frames = hip_decoder.extract_batch(batch)
        H, W = shape[:2]
        for j, k in enumerate(idx):
            want = np.zeros((1024, 1024, 3), np.uint8)
            c8 = (ctypes.c_float * 8)()
            rc = oracle.co_extract(P(cams[k]), W, H, P(want), c8)
            assert status[j] == rc, f"capture {k} {shape}: status {status[j]} vs {rc}"
            if rc:
                assert list(corners[j]) == list(c8), k
                assert (frames[j] == want).all(), f"capture {k}: {(frames[j] != want).sum()} bytes differ"
                ok += 1
            else:
                assert not frames[j].any()
    assert ok >= 4


def test_scan_extract_decode_chain_equals_the_reference_chain(hip_decoder, synth, ref):
    """cimbard_scan_extract_decode / cimbar.cpp's loop for a batch of captures, nothing leaving the device in between: the chunks are the
    reference's (Extractor::extract -> Decoder::decode_fountain with the sharpen verdict), a failed extraction delivers nothing"""
    payload, cams = captures(synth)
    n, h, w = cams.shape[:3]
    blank = np.zeros_like(cams[0])
    batch = np.ascontiguousarray(np.concatenate([cams, blank[None]], 0))       # the last capture has no anchors at all
    hip_decoder.reset_ccm()
    total, chunks, masks, status = hip_decoder.scan_extract_decode_batch(batch, preprocess=-1)
    assert status[n] == 0 and masks[n] == 0 and not chunks[n].any()
    ref.ref_reset_ccm()
    want_total = 0
    for k in range(n):
        ext = np.zeros((1024, 1024, 3), np.uint8)
        rc = ref.ref_extract(P(cams[k]), w, h, P(ext))
        assert status[k] == rc
        r, wchunks, wmask = pyref.ref_decode(ext, rc == 2, 2, 0)
        assert masks[k] == wmask and (chunks[k] == wchunks).all(), k
        want_total += r
    assert total == want_total
    # the caller's explicit choice (cimbar.cpp:186-190, --preprocess 0 | 1) overrides the extractor's verdict for every capture
    for pre in (0, 1):
        hip_decoder.reset_ccm()
        total, chunks, masks, status = hip_decoder.scan_extract_decode_batch(cams, preprocess=pre)
        ref.ref_reset_ccm()
        for k in range(n):
            ext = np.zeros((1024, 1024, 3), np.uint8)
            assert ref.ref_extract(P(cams[k]), w, h, P(ext)) == status[k]
            r, wchunks, wmask = pyref.ref_decode(ext, pre, 2, 0)
            assert masks[k] == wmask and (chunks[k] == wchunks).all(), (pre, k)


@pytest.mark.parametrize("w,h,quad", [(4000, 2600, ((800, 70), (3300, 120), (780, 2500), (3340, 2460))), (5600, 4600, ((600, 90), (4900, 160), (560, 4480), (4960, 4400)))],
                         ids=["9x9-blur", "17x17-blur"])
def test_extract_of_a_capture_that_needs_the_larger_blurs(hip_decoder, synth, oracle, w, h, quad):
    """captures of 2500 px / 4500 px and more on the short side (Scanner's blur unit 9 / 17, Scanner.h:157-159) through the whole of
    Extractor::extract and on through the decoder"""
    payload, frames = F.clean_frames(synth, 1, seed=78)
    cam = np.ascontiguousarray(F.camera_frame(frames[0], width=w, height=h, quad=quad, background=8))
    status, corners, out = hip_decoder.extract_batch(cam[None])
    want = np.zeros((1024, 1024, 3), np.uint8)
    c8 = (ctypes.c_float * 8)()
    st = oracle.co_extract(P(cam), w, h, P(want), c8)
    assert status[0] == st == 1 and list(corners[0]) == list(c8) and (out[0] == want).all()
    hip_decoder.reset_ccm()
    total, chunks, masks, st2 = hip_decoder.scan_extract_decode_batch(cam[None])
    assert st2[0] == 1 and masks[0] == 0xFFF and (chunks[0].reshape(-1) == payload[0]).all()
