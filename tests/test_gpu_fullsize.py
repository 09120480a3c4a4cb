"""GPU, BASELINE-sized inputs: size-independent properties (round trips, idempotence, linearity) AND -- round 6 -- every frame of the 1 024-frame
batches of BASELINE configs[1] / configs[2] against the reference build's own Decoder::decode_fountain (oracle/_ref, /root/reference/src/lib/encoder/
Decoder.h:171-189) on a thread pool: masks and chunk bytes per frame."""
import hashlib

import numpy as np
import pytest
import torch

from libcimbar_amd import framegen, modeb
from tests.refbatch import reference_batch

pytestmark = pytest.mark.gpu


def _assert_equals_reference(frames, chunks, masks):
    ref_chunks, ref_masks = reference_batch(frames.cpu().numpy())
    got_c, got_m = chunks.cpu().numpy(), masks.cpu().numpy().astype(np.uint32)
    assert (got_m == ref_masks).all()
    bad = [k for k in range(got_c.shape[0]) if hashlib.sha256(got_c[k].tobytes()).digest() != hashlib.sha256(ref_chunks[k].tobytes()).digest()]
    assert not bad, f"{len(bad)} frames differ from the reference build, first {bad[:4]}"


def _decode(dec, frames):
    n = frames.shape[0]
    chunks = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=frames.device)
    masks = torch.zeros((n,), dtype=torch.int32, device=frames.device)
    dec.reset_ccm()
    dec.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return chunks, masks


def test_config2_1024_clean_frames_roundtrip(hip_decoder):
    # BASELINE configs[1]: 1024 clean frames; encode -> decode must return every payload byte, every chunk delivered
    dev = torch.device("cuda", 0)
    synth = framegen.FrameSynth(dev)
    payload = framegen.synth_payload(1024, seed=1234, device=dev)
    frames = torch.empty((1024, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
    for lo in range(0, 1024, 64):
        synth.frames_from_payload(payload[lo:lo + 64], out=frames[lo:lo + 64])
    chunks, masks = _decode(hip_decoder, frames)
    assert bool((masks == 0xFFF).all()) and bool((chunks == payload).all())
    # idempotence: the same batch again (CCM now carried in from the previous call) gives the same bytes
    chunks2, masks2 = _decode(hip_decoder, frames)
    assert bool((chunks2 == chunks).all()) and bool((masks2 == masks).all())
    assert not hip_decoder.tap(5, 1024).any()
    # ... and frame by frame what the reference build decodes from the same 1 024 images
    _assert_equals_reference(frames, chunks, masks)


def test_config3_cell_errors_are_all_corrected(hip_decoder):
    # BASELINE configs[2]: 0.8 % of the cells (99) replaced by a different valid tile -> RS corrects every block
    dev = torch.device("cuda", 0)
    synth = framegen.FrameSynth(dev)
    n = 1024   # BASELINE configs[2] at full size
    payload = framegen.synth_payload(n, seed=1234, device=dev)
    tiles = framegen.inject_cell_errors(synth.cell_tiles(payload), n_errors=99, seed=5678)
    frames = torch.empty((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
    for lo in range(0, n, 64):
        synth.render(tiles[lo:lo + 64], out=frames[lo:lo + 64])
    chunks, masks = _decode(hip_decoder, frames)
    assert bool((masks == 0xFFF).all()) and bool((chunks == payload).all())
    assert hip_decoder.tap(4, n).all()          # every RS block decoded
    sym = torch.from_numpy(hip_decoder.tap(1, n).astype(np.int64)).to(dev)
    col = torch.from_numpy(hip_decoder.tap(2, n).astype(np.int64)).to(dev)
    assert bool((col * 16 + sym == tiles).all())   # the per-cell decisions are the substituted tiles: errors really reached RS
    # ... and frame by frame what the reference build (libcorrect's decoder included) makes of the same 1 024 damaged images
    _assert_equals_reference(frames, chunks, masks)


def test_linearity_of_the_payload_path(hip_decoder):
    # RS + interleave + chunking are GF(2)-linear in the payload: decode(frame(a)) ^ decode(frame(b)) == decode(frame(a ^ b))
    dev = torch.device("cuda", 0)
    synth = framegen.FrameSynth(dev)
    a = framegen.synth_payload(8, seed=1, device=dev)
    b = framegen.synth_payload(8, seed=2, device=dev)
    fa, fb, fx = (synth.frames_from_payload(p) for p in (a, b, a ^ b))
    ca, _ = _decode(hip_decoder, fa)
    cb, _ = _decode(hip_decoder, fb)
    cx, mx = _decode(hip_decoder, fx)
    assert bool(((ca ^ cb) == cx).all()) and bool((mx == 0xFFF).all())
