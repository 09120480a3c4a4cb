"""GPU: the encode half (on-device frame synthesiser) against the reference encoder.
libcimbar_amd/framegen.py is byte-identical to Encoder::encode_next (tests/test_framegen_vs_ref.py), so it is the yardstick here;
where oracle/_ref is present the reference build is asked directly as well."""
import numpy as np
import pytest
import torch

from libcimbar_amd import framegen, modeb

pytestmark = pytest.mark.gpu


def test_hip_encoder_matches_reference_layout(hip_decoder, synth):
    payload = framegen.synth_payload(5, seed=77)
    want = synth.frames_from_payload(payload).numpy()
    got = hip_decoder.encode_batch(payload.numpy())
    assert got.shape == want.shape and (got == want).all()
    from oracle import pyref
    if pyref.ref_lib() is not None:
        assert (got[0] == pyref.ref_encode_raw(payload[0].numpy())).all()


def test_encode_decode_roundtrip_on_device(hip_decoder):
    dev = torch.device("cuda", 0)
    n = 128
    payload = framegen.synth_payload(n, seed=4242, device=dev)
    frames = torch.empty((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    hip_decoder.encode_batch_device(payload.data_ptr(), n, frames.data_ptr(), st)
    chunks = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    hip_decoder.reset_ccm()
    hip_decoder.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, st)
    torch.cuda.synchronize()
    assert bool((masks == 0xFFF).all()) and bool((chunks == payload).all())
    # and it is the same picture the torch restatement renders
    synth = framegen.FrameSynth(dev)
    assert bool((frames[:16] == synth.frames_from_payload(payload[:16])).all())
