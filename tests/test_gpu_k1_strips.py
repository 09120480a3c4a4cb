"""The threshold kernel's two instances (round 5): short strips (2 cell rows per wavefront; what a launch that runs alone takes) and tall ones (7; what the
pipelined loop launches for batches of 256 frames and more). Same arithmetic, another partition of the frame over wavefronts -- so every product of K1
(bit plane, undrifted cell means -> colours and colour-correction matrix) and everything downstream must be identical, in every mode, with and
without the sharpen pass's sibling, and against the oracle."""
import numpy as np
import pytest
import torch

from libcimbar_amd import HipDecoder, framegen, geometry
from libcimbar_amd import decoder as D
from oracle import pyref
from oracle.pyref import P
from tests import frames as F
from tests.test_gpu_flood_verify import decoder_with
from tests.test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


def mixed_frames(mode, n=7, seed=505):
    synth_m = framegen.FrameSynth("cpu", mode)
    payload = framegen.synth_payload(n, seed=seed + mode, mode=mode)
    fr = synth_m.frames_from_payload(payload).numpy()
    out = []
    for k, f in enumerate(fr):
        kind = k % 4
        if kind == 1:
            f = F.add_noise(f, 45, k)
        elif kind == 2:
            f = (f.astype(np.float32) * np.array([0.8, 1.0, 0.65], np.float32)).astype(np.uint8)
        elif kind == 3:
            f = F.shift(f, 2, -1)
        out.append(np.ascontiguousarray(f))
    return np.ascontiguousarray(np.stack(out))


@pytest.mark.parametrize("mode", [68, 67, 66, 4, 8])
def test_tall_and_short_strips_give_the_same_everything(mode):
    frames = mixed_frames(mode)
    n = len(frames)
    res = {}
    for pre in (False, True):           # the plain kernel and the sharpening one: both exist in both strip heights
        for which in ("short", "tall"):
            dec = decoder_with({"CIMBAR_HIP_K1_STRIPS": which}, mode)
            total, chunks, masks = dec.decode_batch(frames, should_preprocess=pre)
            res[which] = (total, chunks.copy(), masks.copy(), dec.tap(D.TAP_BITPLANE, n).copy(), dec.tap(D.TAP_SYMBOLS, n).copy(),
                          dec.tap(D.TAP_COLORS, n).copy(), dec.tap(D.TAP_CCM, n).copy(), dec.tap(D.TAP_DRIFT, n).copy())
            dec.close()
        a, b = res["short"], res["tall"]
        assert a[0] == b[0]
        for name, x, y in zip(("chunks", "masks", "bit plane", "symbols", "colours", "ccm", "drift"), a[1:], b[1:]):
            assert x.tobytes() == y.tobytes(), f"mode {mode}, preprocess {pre}: {name} differ between the two strip heights"
        assert (a[2] != 0).any()


def test_tall_strips_against_the_oracle(synth):
    dec = decoder_with({"CIMBAR_HIP_K1_STRIPS": "tall"})
    items = F.distorted_set(synth, seed=91)[:9]
    check_against_oracle(dec, [f for _, f in items], names=[n for n, _ in items])
    check_against_oracle(dec, [f for _, f in items[:6]], pre=1, names=[n for n, _ in items[:6]])
    dec.close()


def test_the_pipelined_loop_switches_instance_with_the_batch_size(synth):
    """256 frames per step through the pipelined entry point (tall strips) equal the ordinary call (short strips) frame for frame; a 255-frame step
    (short strips again) as well"""
    dev = torch.device("cuda", 0)
    payload = framegen.synth_payload(256, seed=77, device=dev)
    dec = HipDecoder(0)
    frames = torch.empty((256, 1024, 1024, 3), dtype=torch.uint8, device=dev)
    dec.encode_batch_device(payload.data_ptr(), 256, frames.data_ptr())
    st = torch.cuda.current_stream(dev)
    want_c, want_m = torch.zeros((256, 7500), dtype=torch.uint8, device=dev), torch.zeros(256, dtype=torch.int32, device=dev)
    dec.decode_batch_device(frames.data_ptr(), 256, want_c.data_ptr(), want_m.data_ptr(), False, 2, st.cuda_stream)
    torch.cuda.synchronize(dev)
    assert bool((want_c == payload).all().item()) and bool((want_m == 0xFFF).all().item())
    for n in (256, 255):
        got_c, got_m = torch.zeros((n, 7500), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        dec.decode_batch_pipelined(frames.data_ptr(), n, got_c.data_ptr(), got_m.data_ptr(), False, 2, st.cuda_stream)
        dec.pipeline_wait(st.cuda_stream)
        torch.cuda.synchronize(dev)
        assert bool((got_c == want_c[:n]).all().item()) and bool((got_m == want_m[:n]).all().item()), n
        plane = dec.tap(D.TAP_BITPLANE, n)[:4].copy()          # (the tap holds the last batch: n frames)
        tmp_c, tmp_m = torch.zeros((4, 7500), dtype=torch.uint8, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
        dec.decode_batch_device(frames.data_ptr(), 4, tmp_c.data_ptr(), tmp_m.data_ptr(), False, 2, st.cuda_stream)
        torch.cuda.synchronize(dev)
        assert (plane == dec.tap(D.TAP_BITPLANE, 4)).all()
    dec.close()


@pytest.mark.parametrize("strips", ["short", "tall"])
@pytest.mark.parametrize("MODE", [68, 67, 66])
def test_bitplane_of_images_that_are_busy_at_their_borders(MODE, strips, monkeypatch):
    """a frame's margin is one flat colour, so the frames above cannot tell BORDER_REPLICATE (box mean) from BORDER_REFLECT_101 (sharpen filter) from
    anything else at the image's edges. Noise, a diagonal ramp and single bright / dark edge columns and rows can: both threshold variants, both
    strip heights, every bit of the plane. (Mode 66's rows end inside the last lane of the threshold pass: four of its twelve pixel slots exist.)"""
    monkeypatch.setenv("CIMBAR_HIP_K1_STRIPS", strips)
    GEO = geometry.for_mode(MODE)
    h, w = GEO.IMG_H, GEO.IMG_W
    imgs = F.border_images(h, w, 66 + MODE)
    O = pyref.oracle_lib(MODE)
    d = D.HipDecoder(0, MODE)
    try:
        for pre in (0, 1):
            d.decode_batch(np.stack(imgs), should_preprocess=pre)
            got = d.tap(D.TAP_BITPLANE, len(imgs))
            for k, im in enumerate(imgs):
                want = np.zeros(w * h // 8, np.uint8)
                O.co_threshold_bitplane(P(np.ascontiguousarray(im)), w, h, pre, P(want))
                bad = np.flatnonzero(got[k] != want)
                assert bad.size == 0, f"pre={pre} image {k}: {bad.size} bitplane bytes differ, first at row {bad[0] // (w // 8)} byte {bad[0] % (w // 8)}"
    finally:
        d.close()
