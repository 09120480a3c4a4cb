"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs. Bit-exact everywhere."""
import numpy as np
import pytest

from libcimbar_amd import decoder as D
from libcimbar_amd import modeb
from oracle import pyref
from tests import frames as F

pytestmark = pytest.mark.gpu


def oracle_batch(frames, pre=0, cc=2, ccm=None):
    """decode frames in order with the oracle, carrying the CCM like one reference thread would."""
    ccm = ccm or pyref.CoCcm()
    outs = []
    for fr in frames:
        r, chunks, mask, ccm = pyref.oracle_decode(fr, pre, cc, ccm)
        sym, col, pos = pyref.oracle_stage()
        m = np.array(list(ccm.m), np.float32)
        outs.append(dict(r=r, chunks=chunks.copy(), mask=mask, sym=sym, col=col, pos=pos, ccm=m, active=ccm.active))
    return outs


def check_against_oracle(dec, frames, pre=0, cc=2, names=None):
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    dec.reset_ccm()
    total, chunks, masks = dec.decode_batch(frames, should_preprocess=pre, color_correction=cc)
    want = oracle_batch(frames, pre, cc)
    sym = dec.tap(D.TAP_SYMBOLS, n)
    col = dec.tap(D.TAP_COLORS, n)
    drift = dec.tap(D.TAP_DRIFT, n)
    ccm = dec.tap(D.TAP_CCM, n)
    xy = modeb.cell_positions()
    for k in range(n):
        tag = names[k] if names else str(k)
        w = want[k]
        assert (sym[k] == w["sym"]).all(), f"{tag}: symbols differ in {(sym[k] != w['sym']).sum()} cells"
        assert (xy + drift[k].astype(np.int32) == w["pos"]).all(), f"{tag}: colour positions differ"
        assert ccm[k, :9].tobytes() == w["ccm"].tobytes() or not w["active"], f"{tag}: CCM differs {ccm[k]} vs {w['ccm']}"
        assert bool(ccm[k, 9]) == bool(w["active"]), f"{tag}: CCM active flag"
        assert (col[k] == w["col"]).all(), f"{tag}: colours differ in {(col[k] != w['col']).sum()} cells"
        assert masks[k] == w["mask"], f"{tag}: mask {masks[k]:#x} vs {w['mask']:#x}"
        assert (chunks[k] == w["chunks"]).all(), f"{tag}: chunk bytes differ"
    assert total == sum(w["r"] for w in want)
    return chunks, masks, want


def test_clean_batch_bit_exact(hip_decoder, synth):
    payload, frames = F.clean_frames(synth, 6, seed=1234)
    chunks, masks, _ = check_against_oracle(hip_decoder, list(frames))
    assert (masks == 0xFFF).all()
    assert (chunks.reshape(6, -1) == payload).all()
    assert not hip_decoder.tap(D.TAP_FLOOD, 6).any(), "clean frames must take the parallel path"


def test_bitplane_matches_oracle(hip_decoder, synth, oracle):
    _, frames = F.clean_frames(synth, 2, seed=9)
    frames = [frames[0], F.add_noise(frames[1], 60, 5)]
    for pre in (0, 1):
        hip_decoder.decode_batch(np.stack(frames), should_preprocess=pre)
        got = hip_decoder.tap(D.TAP_BITPLANE, 2)
        for k in range(2):
            want = np.zeros(131072, np.uint8)
            oracle.co_threshold_bitplane(pyref.P(np.ascontiguousarray(frames[k])), 1024, 1024, pre, pyref.P(want))
            assert (got[k] == want).all(), f"pre={pre} frame {k}: {(got[k] != want).sum()} bitplane bytes differ"


def test_tile_substitution_errors_are_corrected(hip_decoder, synth):
    payload, frames = F.tile_error_frames(synth, 4, seed=4321, n_errors=99)
    chunks, masks, _ = check_against_oracle(hip_decoder, list(frames))
    assert (masks == 0xFFF).all()
    assert (chunks.reshape(4, -1) == payload).all()
    rs_ok = hip_decoder.tap(D.TAP_RS_OK, 4)
    assert rs_ok.all()


def test_distorted_frames_match_oracle(hip_decoder, synth):
    items = F.distorted_set(synth)
    names = [n for n, _ in items]
    frames = [f for _, f in items]
    check_against_oracle(hip_decoder, frames, names=names)
    flood = hip_decoder.tap(D.TAP_FLOOD, len(frames))
    assert flood[names.index("shift+2+1")] == 1, "a shifted frame must take the exact flood path"


def test_distorted_frames_with_preprocess(hip_decoder, synth):
    items = F.distorted_set(synth, seed=78)[:8]
    check_against_oracle(hip_decoder, [f for _, f in items], pre=1, names=[n for n, _ in items])


@pytest.mark.parametrize("cc", [0, 1, 2])
def test_color_correction_modes_and_carry(hip_decoder, synth, cc):
    # frame 1 is wiped so hard that no symbol chunk survives: with cc=2 it must reuse frame 0's matrix (CimbDecoder.cpp:69-85)
    _, frames = F.clean_frames(synth, 3, seed=31)
    frames = [frames[0], F.blank_region(frames[1], 0, 1024, 0, 1024, value=0), F.add_noise(frames[2], 50, 9)]
    check_against_oracle(hip_decoder, frames, cc=cc)


def test_decode_frame_and_fountain_surface(hip_decoder, synth):
    payload, frames = F.clean_frames(synth, 1, seed=3)
    hip_decoder.reset_ccm()
    good, chunks, mask = hip_decoder.decode_frame(frames[0])
    assert good == 7500 and mask == 0xFFF and (chunks.reshape(-1) == payload[0]).all()

    class Sink:
        def __init__(self, cs): self.cs, self.got = cs, []
        def chunk_size(self): return self.cs
        def write(self, b): self.got.append(b)
    s = Sink(625)
    assert hip_decoder.decode_fountain(frames[0], s) == 7500
    assert b"".join(s.got) == payload[0].tobytes()
    s2 = Sink(600)   # Decoder.h:180-185: chunk-size mismatch -> decode to a null stream, still report the bytes
    assert hip_decoder.decode_fountain(frames[0], s2) == 7500 and not s2.got


def test_bad_arguments(hip_decoder):
    with pytest.raises(D.CimbarHipError):
        hip_decoder.decode_batch(np.zeros((2, 512, 512, 3), np.uint8))          # the batch entry points take exact-size frames only
    # (decode_frame follows CimbReader's constructor for other sizes: tests/test_padded_frames.py)


@pytest.mark.parametrize("pre", [0, 1])
def test_plain_decode_matches_oracle(hip_decoder, synth, pre):
    """cimbar_hip_decode_plain_batch = Decoder::decode (--no-fountain): RS outputs back to back, failed blocks zeroed, and with
    color_correction 2 no matrix of its own (no fountain header reaches the reader): frames reuse the carried one."""
    names, frames = zip(*F.distorted_set(synth))
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    hip_decoder.reset_ccm()
    total, data, ok = hip_decoder.decode_plain_batch(frames, should_preprocess=pre)
    assert total == 7500 * n
    ccm = pyref.CoCcm()
    for k in range(n):
        r, want, wok, ccm = pyref.oracle_decode_plain(frames[k], pre, 2, ccm)
        assert r == 7500
        assert (ok[k] == wok).all(), f"{names[k]}: RS block flags differ"
        assert (data[k] == want).all(), f"{names[k]}: bytes differ"
    # a carried matrix (left by a fountain decode on this context) is used by the plain path, like the reference's thread_local
    _, clean = F.clean_frames(synth, 1, seed=8)
    tinted = F.add_noise(clean[0], 30, 1)
    hip_decoder.reset_ccm()
    hip_decoder.decode_frame(tinted)                       # leaves a CCM behind
    assert hip_decoder.get_ccm()[0]
    _, d1, ok1 = hip_decoder.decode_plain_batch(tinted[None])
    occ = pyref.CoCcm()
    pyref.oracle_decode(tinted, 0, 2, occ)
    _, w1, wok1, _ = pyref.oracle_decode_plain(tinted, 0, 2, occ)
    assert (d1[0] == w1).all() and (ok1[0] == wok1).all()
