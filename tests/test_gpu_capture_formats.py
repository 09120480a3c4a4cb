"""GPU: the capture path with the `format` argument of the reference's C ABI (cimbard_scan_extract_decode, cimbar_recv_js.h:17; get_rgb,
cimbar_recv_js.cpp:94-120): RGBA, NV12 and the three-plane 4:2:0 layout are converted inside the kernels that read the capture (X1 gray +
blur, X4 warp). Checked stage by stage against the oracle (conversion restated in oracle/cimbar_oracle_extract.c, pinned to the cv-shim and to
the reference's chain in tests/test_capture_formats.py) and end to end against the reference's own cimbard_scan_extract_decode."""
import ctypes

import numpy as np
import pytest
import torch

from libcimbar_amd import decoder as D
from oracle import pyref
from oracle.pyref import P
from tests import capture_formats as CF
from tests import frames as F
from tests.test_capture_formats import reference_capture_chain
from tests.test_gpu_flood_verify import decoder_with
from tests.test_oracle_vs_ref import CAMERA_CASES

pytestmark = pytest.mark.gpu
FMTS = CF.FORMATS


def to_rgb(oracle, img, w, h, fmt):
    out = np.zeros((h, w, 3), np.uint8)
    assert oracle.co_capture_to_rgb(P(img), w, h, fmt, P(out)) == 0
    return out


def camera_set(synth, fmt, size=(1920, 1080)):
    payload, frames = F.clean_frames(synth, len(CAMERA_CASES), seed=90)
    cams = [CF.rgb_to_format(F.camera_frame(frames[k], width=size[0], height=size[1], quad=q, background=bg, blur=bl), fmt) for k, (bg, q, bl) in enumerate(CAMERA_CASES)]
    return payload, np.ascontiguousarray(np.stack(cams))


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("size", [(1920, 1080), (2608, 1700), (1282, 978), (64, 40), (4000, 2600)],
                         ids=["3x3-streaming", "5x5-streaming", "3x3-tiled-odd-width", "tiny", "9x9-tiled"])
def test_scan_preprocess_in_every_format(hip_decoder, oracle, fmt, size):
    w, h = size
    rng = np.random.default_rng(w + fmt)
    img = rng.integers(0, 256, (1, CF.capture_bytes(w, h, fmt)), dtype=np.uint8)
    img[0, : w * (h // 2)] //= 3                                   # a dark half so that Otsu has something to find
    got, thr = hip_decoder.scan_preprocess(img, size=size, fmt=fmt)
    want = np.zeros((h, w), np.uint8)
    rgb = to_rgb(oracle, img[0], w, h, fmt)
    assert thr[0] == oracle.co_scan_preprocess(P(rgb), w, h, P(want))
    assert (got[0] == want).all(), f"{(got[0] != want).sum()} pixels differ"


@pytest.mark.parametrize("fmt", FMTS)
def test_extract_batch_in_every_format(hip_decoder, synth, oracle, fmt):
    """Extractor::extract on captures in `fmt`: status, corners and the deskewed frame == the oracle's get_rgb + Extractor restatement; with
    corners that push part of the frame outside the capture (the warp's border path) as a second deskew"""
    _, cams = camera_set(synth, fmt)
    w, h = 1920, 1080
    status, corners, frames = hip_decoder.extract_batch(cams, size=(w, h), fmt=fmt)
    for k in range(len(cams)):
        want = np.zeros((1024, 1024, 3), np.uint8)
        c8 = (ctypes.c_float * 8)()
        rc = oracle.co_extract_fmt(P(cams[k]), w, h, fmt, P(want), c8)
        assert status[k] == rc and rc in (1, 2)
        assert list(corners[k]) == list(c8)
        assert (frames[k] == want).all(), f"capture {k}: {(frames[k] != want).sum()} bytes differ"
    far = corners.copy()
    far[:, 0] -= 500; far[:, 1] -= 300; far[:, 6] += 450; far[:, 7] += 200
    desk = hip_decoder.deskew_batch(cams, far, size=(w, h), fmt=fmt)
    for k in range(len(cams)):
        want = np.zeros((1024, 1024, 3), np.uint8)
        rgb = to_rgb(oracle, cams[k], w, h, fmt)
        oracle.co_deskew(P(rgb), w, h, far[k].ctypes.data_as(ctypes.POINTER(ctypes.c_float)), P(want))
        assert (desk[k] == want).all(), f"capture {k}: {(desk[k] != want).sum()} bytes differ"


@pytest.mark.parametrize("fmt", FMTS)
def test_chain_equals_the_references_cimbard_scan_extract_decode(hip_decoder, synth, ref, fmt):
    """cimbar_hip_scan_extract_decode_batch_fmt(..., preprocess 1, colour correction 2) against the reference's own C entry point, capture by
    capture in order (both sides carry their colour-correction matrix from capture to capture), host buffers and device buffers"""
    payload, cams = camera_set(synth, fmt)
    w, h = 1920, 1080
    blank = np.full((1, cams.shape[1]), 16 if fmt in (12, 420) else 0, np.uint8)
    batch = np.ascontiguousarray(np.concatenate([cams, blank], 0))
    hip_decoder.reset_ccm()
    total, chunks, masks, status = hip_decoder.scan_extract_decode_batch(batch, preprocess=1, size=(w, h), fmt=fmt)
    ref.ref_reset_ccm()
    want_total = 0
    for k in range(len(batch)):
        r, buf = reference_capture_chain(ref, batch[k], w, h, fmt)
        if r == -3:
            assert status[k] == 0 and masks[k] == 0 and not chunks[k].any()
            continue
        packed = np.concatenate([chunks[k][j] for j in range(12) if masks[k] >> j & 1] + [np.zeros(0, np.uint8)])
        assert packed.size == r and (packed == buf[:r]).all(), k
        want_total += r
    assert total == want_total >= 7500
    # the same batch without leaving the device
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    d_in = torch.from_numpy(batch).to(dev)
    n = len(batch)
    d_chunks = torch.zeros((n, 7500), dtype=torch.uint8, device=dev)
    d_masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    d_status = torch.zeros((n,), dtype=torch.int32, device=dev)
    hip_decoder.reset_ccm()
    hip_decoder.scan_extract_decode_device(d_in.data_ptr(), w, h, n, d_chunks.data_ptr(), d_masks.data_ptr(), d_status.data_ptr(), preprocess=1, stream=st, fmt=fmt)
    torch.cuda.synchronize()
    assert (d_chunks.cpu().numpy() == chunks.reshape(n, -1)).all() and (d_masks.cpu().numpy().astype(np.uint32) == masks).all()
    assert (d_status.cpu().numpy() == status).all()


def test_format_argument_edges(hip_decoder):
    lib = D.load_library()
    # 4:2:0 layouts cannot hold an odd width or height (cv::cvtColor asserts it; the reference would throw): EDIM, and capture_bytes says 0
    assert lib.cimbar_hip_capture_bytes(63, 64, 12) == 0 and lib.cimbar_hip_capture_bytes(64, 63, 420) == 0
    buf = np.zeros(64 * 64 * 4, np.uint8)
    out = np.zeros(64 * 64, np.uint8)
    for fmt in (12, 420):
        assert lib.cimbar_hip_scan_preprocess_fmt(hip_decoder._ctx, buf.ctypes.data, 63, 64, fmt, 1, D.MEM_HOST, out.ctypes.data, None, D.MEM_HOST, None) == -2
    # `format <= 0` is 3 (cimbar_recv_js.cpp:150-151), and so is any value get_rgb's default: lets through
    rgb = np.random.default_rng(5).integers(0, 256, (1, 48, 64, 3), dtype=np.uint8)
    want, thr = hip_decoder.scan_preprocess(rgb)
    for fmt in (0, -7, 3, 5):
        got, thr2 = hip_decoder.scan_preprocess(rgb.reshape(1, -1), size=(64, 48), fmt=fmt)
        assert (got == want).all() and thr2[0] == thr[0]


@pytest.mark.parametrize("fmt", [12, 420])
def test_two_pass_warp_in_several_passes(hip_decoder, synth, oracle, fmt):
    """4:2:0 captures are converted to RGB once per source pixel ahead of the warp, a scratch allowance's worth of captures at a time: with 16 MB of
    allowance a batch of 1080p captures (6.2 MB each) goes two per pass with an odd one left -- the same frames as in one pass, as with the
    conversion inside the warp kernel (CIMBAR_HIP_WARP_TWOPASS=0), and as the oracle's"""
    _, cams = camera_set(synth, fmt)
    cams = np.ascontiguousarray(np.concatenate([cams, cams[:3]]))
    assert len(cams) % 2 == 1 and len(cams) >= 5
    w, h = 1920, 1080
    status, corners, frames = hip_decoder.extract_batch(cams, size=(w, h), fmt=fmt)
    for env in ({"CIMBAR_HIP_WARP_SCRATCH_MB": "16"}, {"CIMBAR_HIP_WARP_TWOPASS": "0"}):
        d = decoder_with(env)
        try:
            s2, c2, f2 = d.extract_batch(cams, size=(w, h), fmt=fmt)
        finally:
            d.close()
        assert (s2 == status).all() and (c2 == corners).all() and (f2 == frames).all(), env
    want = np.zeros((1024, 1024, 3), np.uint8)
    c8 = (ctypes.c_float * 8)()
    assert oracle.co_extract_fmt(P(cams[-1]), w, h, fmt, P(want), c8) == status[-1]
    assert (frames[-1] == want).all()
