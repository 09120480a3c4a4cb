"""GPU: the batch-parallel flood's certificate (k_flood_wave: "no tie-break of the reference's heap could have changed this frame's result")
checked against the exact replay on ~100 000 distorted frames in three geometries, GPU vs GPU, through the library's own verify mode
(CIMBAR_HIP_FLOOD_VERIFY=1: every certified frame is replayed exactly and compared cell by cell, CIMBAR_HIP_TAP_FLOOD_VERIFY) -- and the
exact replay (k_flood3) against the oracle."""
import os

import numpy as np
import pytest
import torch

from libcimbar_amd import decoder as D
from libcimbar_amd import framegen, geometry
from tools import extractbench
from oracle import pyref
from tests import frames as F

pytestmark = pytest.mark.gpu


def decoder_with(env, mode=68, lib_path=None):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return D.HipDecoder(0, mode, lib_path=lib_path)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def distort_group(fr, g, dev):
    """one random recipe applied to a group of device frames (n, h, w, 3) uint8: rigid shift, wipe, noise, tear, rescale, or a sub-pixel
    (bilinear) warp -- and mixes of them"""
    n, h, w, _ = fr.shape
    out = torch.roll(fr, shifts=(int(g.integers(-4, 5)), int(g.integers(-4, 5))), dims=(1, 2))
    kind = int(g.integers(0, 8))
    if kind == 1:                                   # a wiped block
        y0, x0 = int(g.integers(0, h - 60)), int(g.integers(0, w - 60))
        out = out.clone()
        out[:, y0:y0 + int(g.integers(20, 400)), x0:x0 + int(g.integers(20, 400))] = int(g.integers(0, 256))
    elif kind == 2:                                 # pixel noise
        noise = torch.randn(out.shape, device=dev, dtype=torch.float16) * float(g.integers(3, 50))
        out = (out.to(torch.float16) + noise).round().clamp(0, 255).to(torch.uint8)
    elif kind == 3:                                 # a tear: two different shifts
        cut = int(g.integers(100, h - 100))
        out = torch.cat([torch.roll(fr, shifts=(1, 0), dims=(1, 2))[:, :cut], torch.roll(fr, shifts=(0, 1), dims=(1, 2))[:, cut:]], 1)
    elif kind == 4:                                 # up to the drift limit
        out = torch.roll(fr, shifts=(int(g.integers(-7, 8)), int(g.integers(-7, 8))), dims=(1, 2))
    elif kind in (5, 6):                            # rescale about the centre / small rotation, bilinear (what a deskewed capture looks like)
        s = 1.0 + float(g.uniform(-0.008, 0.008)) if kind == 5 else 1.0
        a = float(g.uniform(-0.004, 0.004)) if kind == 6 else 0.0
        theta = torch.tensor([[s * np.cos(a), -s * np.sin(a) * h / w, float(g.uniform(-0.004, 0.004))],
                              [s * np.sin(a) * w / h, s * np.cos(a), float(g.uniform(-0.004, 0.004))]], dtype=torch.float32, device=dev)
        grid = torch.nn.functional.affine_grid(theta[None].expand(n, -1, -1), (n, 3, h, w), align_corners=False)
        src = out.permute(0, 3, 1, 2).to(torch.float32)
        out = torch.nn.functional.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=False)
        out = out.permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8)
    elif kind == 7:                                 # random 9x9 patches
        out = out.clone()
        for _ in range(int(g.integers(1, 30))):
            y, x = int(g.integers(8, h - 20)), int(g.integers(8, w - 20))
            out[:, y:y + 9, x:x + 9] = torch.randint(0, 256, (9, 9, 3), device=dev, dtype=torch.uint8)
    return out.contiguous()


@pytest.mark.parametrize("mode,batches", [(68, 50), (67, 26), (66, 24)])
def test_certificate_holds_on_100k_distorted_frames(mode, batches):
    """~100 k frames over the three geometries (1024 per batch): every frame k_flood_wave certifies equals its own exact replay"""
    dev = torch.device("cuda", 0)
    geo = geometry.for_mode(mode)
    dec = decoder_with({"CIMBAR_HIP_FLOOD_VERIFY": "1"}, mode)
    st = torch.cuda.current_stream(dev).cuda_stream
    n, group = 1024, 128
    g = np.random.default_rng(9000 + mode)
    frames = torch.empty((n, geo.IMG_H, geo.IMG_W, 3), dtype=torch.uint8, device=dev)
    chunks = torch.zeros((n, geo.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    certified = exact = bad = 0
    for b in range(batches):
        payload = framegen.synth_payload(n, seed=100 * mode + b, device=dev, mode=mode)
        dec.encode_batch_device(payload.data_ptr(), n, frames.data_ptr(), st)
        torch.cuda.synchronize(dev)
        dist = torch.cat([distort_group(frames[k:k + group], g, dev) for k in range(0, n, group)], 0).contiguous()
        dec.decode_batch_device(dist.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, st)
        torch.cuda.synchronize(dev)
        path = dec.tap(D.TAP_FLOOD_PATH, n)
        ver = dec.tap(D.TAP_FLOOD_VERIFY, n)
        assert ((ver != 0xFFFFFFFF) == (path == 2)).all(), "exactly the certified frames are verified"
        wrong = np.flatnonzero((path == 2) & (ver != 0))
        assert wrong.size == 0, f"mode {mode} batch {b}: certified frames {wrong[:8]} differ from their exact replay in {ver[wrong[:8]]} cells"
        certified += int((path == 2).sum())
        exact += int((path == 1).sum())
        del dist
    dec.close()
    print(f"mode {mode}: {batches * n} frames, {certified} certified and verified, {exact} exact, 0 differed")
    assert certified >= batches * n // 8, f"too few frames certified to call this a test: {certified}"


def camera_like(synth, n, seed):
    """deskewed captures: clean frames drawn into 1080p canvases (bilinear) and warped back by the oracle's Extractor"""
    payload, fr = F.clean_frames(synth, max(n, 4), seed=seed)     # (make_captures deals its quads round robin: at least one frame per quad)
    caps = extractbench.make_captures(torch.from_numpy(fr)).numpy()
    L = pyref.oracle_lib()
    out = []
    for k in range(n):
        frame = np.zeros((1024, 1024, 3), np.uint8)
        c8 = np.zeros(8, np.float32)
        assert L.co_extract(pyref.P(np.ascontiguousarray(caps[k])), 1920, 1080, pyref.P(frame), pyref.P(c8)) > 0
        out.append(frame)
    return out


@pytest.mark.parametrize("dense", [0, 1])
def test_exact_replay_kernel_against_the_oracle(synth, dense):
    """(dense = 1: the eight-frames-per-CU instance -- visited bits in LDS, offered priorities in global memory, 4 160-slot LDS heap, so the
    shifted frames' heaps also cross into the spill area.) k_flood3 (the exact replay; the earlier generations k_flood / k_flood2 it was first checked against are gone) on shifted, noisy,
    rescaled, pure-noise and camera-like frames, every flagged frame through the exact replay, with and without the sharpening threshold:
    symbols, drifted positions, chunks and masks against the oracle's std::priority_queue-order restatement"""
    from libcimbar_amd import modeb
    payload, fr = F.clean_frames(synth, 6, seed=606)
    g = np.random.default_rng(66)
    frames = [F.shift(fr[0], 2, 1), F.add_noise(F.shift(fr[1], -3, 2), 40, 7), F.rescale(fr[2], 6), g.integers(0, 256, (1024, 1024, 3), dtype=np.uint8),
              F.add_noise(fr[3], 90, 3), F.rescale(fr[4], 11), F.shift(fr[5], -7, 7)]
    frames += camera_like(synth, 4, seed=607)
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    xy = modeb.cell_positions()
    dec = decoder_with({"CIMBAR_HIP_FLOOD_WAVE": "0", "CIMBAR_HIP_FLOOD_DENSE": str(dense)})
    for pre in (0, 1):
        dec.reset_ccm()
        total, chunks, masks = dec.decode_batch(frames, should_preprocess=pre)
        sym, drift, path = dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_DRIFT, n), dec.tap(D.TAP_FLOOD_PATH, n)
        assert (path[:7] == 1).all()
        ccm = pyref.CoCcm()
        for k in range(n):
            r, wchunks, wmask, ccm = pyref.oracle_decode(frames[k], pre, 2, ccm)
            wsym, wcol, wpos = pyref.oracle_stage()
            assert (sym[k] == wsym).all(), f"pre {pre} frame {k}: {(sym[k] != wsym).sum()} symbols differ from the oracle"
            if path[k]:
                assert (xy + drift[k].astype(np.int32) == wpos).all(), f"pre {pre} frame {k}: drifted positions differ"
            assert masks[k] == wmask and (chunks[k] == wchunks).all(), (pre, k)
    dec.close()


def test_dense_replay_hands_frames_out_when_there_are_more_frames_than_workgroups(synth):
    """CIMBAR_HIP_FLOOD_DENSE_GRID=8 with 30 shifted / rescaled / noisy frames: the launch has 8 workgroups, so 22 frames are handed out through the
    global counter pair (FloodScratch::next, zeroed by the host in front of the launch). Twice on the same context (the second launch must start
    from a zero counter again), every frame against the oracle."""
    payload, fr = F.clean_frames(synth, 6, seed=616)
    g = np.random.default_rng(617)
    frames = []
    for k in range(30):
        base = fr[k % 6]
        kind = k % 3
        if kind == 0:
            frames.append(F.shift(base, int(g.integers(-6, 7)), int(g.integers(-6, 7)) or 1))
        elif kind == 1:
            frames.append(F.add_noise(F.shift(base, int(g.integers(-3, 4)) or 2, int(g.integers(-3, 4))), 35, 100 + k))
        else:
            frames.append(F.rescale(base, int(g.integers(3, 12))))
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    dec = decoder_with({"CIMBAR_HIP_FLOOD_WAVE": "0", "CIMBAR_HIP_FLOOD_DENSE": "1", "CIMBAR_HIP_FLOOD_DENSE_GRID": "8"})
    want = []
    ccm = pyref.CoCcm()
    for k in range(n):
        r, wchunks, wmask, ccm = pyref.oracle_decode(frames[k], 0, 2, ccm)
        want.append((wchunks.copy(), wmask, pyref.oracle_stage()[0].copy()))
    for rep in range(2):
        dec.reset_ccm()
        total, chunks, masks = dec.decode_batch(frames)
        sym, path = dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_FLOOD_PATH, n)
        assert (path == 1).sum() >= 24, "most frames were meant to take the exact replay (8 workgroups: the rest are handed out by the counter)"
        for k in range(n):
            wchunks, wmask, wsym = want[k]
            assert (sym[k] == wsym).all(), f"launch {rep} frame {k}: {(sym[k] != wsym).sum()} symbols differ from the oracle"
            assert masks[k] == wmask and (chunks[k] == wchunks).all(), (rep, k)
    dec.close()


@pytest.mark.parametrize("dense", [0, 1])
def test_exact_replay_kernel_with_the_heap_spilling(synth, dense):
    from libcimbar_amd import build as hipbuild
    from tests.test_gpu_flood import check, flood_frames
    dec = decoder_with({"CIMBAR_HIP_FLOOD_WAVE": "0", "CIMBAR_HIP_FLOOD_DENSE": str(dense)}, lib_path=hipbuild.OUT_SPILLTEST)
    frames, names = flood_frames(synth)
    frames = frames + camera_like(synth, 1, seed=608)
    check(dec, frames, names + ["camera-like"])
    dec.close()


def test_the_certifying_pass_is_skipped_where_it_achieves_nothing_and_nothing_changes(synth):
    """enqueue()'s scheduling heuristic: after a batch in which k_flood_wave certified (almost) none of the frames it was given, the following
    batches go straight to the exact replay and every sixteenth probes again; a stream it does certify keeps it. Whatever it decides, chunks,
    masks, symbols and drift are the ones CIMBAR_HIP_FLOOD_WAVE_ADAPT=0 produces."""
    dev = torch.device("cuda", 0)
    n = 256
    payload, fr = F.clean_frames(synth, 4, seed=611)
    g = np.random.default_rng(612)
    base = torch.from_numpy(np.ascontiguousarray(fr)).to(dev)
    base = base[torch.arange(n, device=dev) % 4].contiguous()
    # what a deskewed capture looks like: a bilinear rescale by half a percent plus pixel noise (the certifier gives up on these in its first rounds)
    h, w = base.shape[1:3]
    theta = torch.tensor([[1.005, 0.0, 0.002], [0.0, 1.005, -0.001]], dtype=torch.float32, device=dev)
    noisy = torch.empty_like(base)
    for k in range(0, n, 64):
        grid = torch.nn.functional.affine_grid(theta[None].expand(64, -1, -1), (64, 3, h, w), align_corners=False)
        src = base[k:k + 64].permute(0, 3, 1, 2).to(torch.float32)
        out = torch.nn.functional.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=False).permute(0, 2, 3, 1)
        noisy[k:k + 64] = (out + torch.randn(out.shape, device=dev) * 20).round().clamp(0, 255).to(torch.uint8)
        del grid, src, out
    noisy = noisy.contiguous()
    shifted = torch.roll(base, shifts=(2, 1), dims=(1, 2)).contiguous()
    st = torch.cuda.current_stream(dev).cuda_stream
    outs = {}
    for name, env in (("adaptive", {}), ("always", {"CIMBAR_HIP_FLOOD_WAVE_ADAPT": "0"})):
        dec = decoder_with(env)
        chunks = torch.zeros((n, dec.geo.FRAME_BYTES), dtype=torch.uint8, device=dev)
        masks = torch.zeros((n,), dtype=torch.int32, device=dev)
        rows = []
        # 20 noisy batches (nothing certifies), then shifted ones (everything does): the probe after the skipped stretch must bring the pass back
        for b in range(40):
            src = noisy if b < 20 else shifted
            dec.reset_ccm()
            dec.decode_batch_device(src.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, st)
            torch.cuda.synchronize(dev)
            path = dec.tap(D.TAP_FLOOD_PATH, n)
            rows.append((chunks.cpu().numpy().copy(), masks.cpu().numpy().copy(), dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_DRIFT, n), path))
        outs[name] = rows
        dec.close()
    for b in range(40):
        a, r = outs["adaptive"][b], outs["always"][b]
        assert (a[0] == r[0]).all() and (a[1] == r[1]).all() and (a[2] == r[2]).all(), f"batch {b}: results depend on the scheduling heuristic"
        flagged = r[4] != 0
        assert (a[3][flagged] == r[3][flagged]).all(), f"batch {b}: drift differs"
    assert all((outs["always"][b][4] == 1).mean() > 0.9 for b in range(20)), "the noisy frames were meant to defeat the certifier"
    assert all((outs["always"][b][4] == 2).all() for b in range(20, 40))
    certified = [int((outs["adaptive"][b][4] == 2).sum()) for b in range(40)]
    assert sum(certified[20:]) > 0 and certified[-1] == n, f"the pass never came back for the frames it certifies: {certified}"
    assert sum(1 for b in range(20, 40) if certified[b] == n) >= 3
