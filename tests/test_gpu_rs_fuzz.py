"""GPU: Reed-Solomon decode fuzzed through the whole pipeline. Frames are rendered from hand-made symbol / colour streams
(valid RS(155,125) codewords with 0..22 injected byte errors per block, at random and at edge positions), so every one of the
60 blocks of a frame exercises K3 with a known error pattern; per-block outcome and bytes are compared with the oracle's
literal libcorrect restatement -- including the >15-error regime where libcorrect may "succeed" with wrong data."""
import numpy as np
import pytest
import torch

from libcimbar_amd import decoder as D
from libcimbar_amd import framegen, modeb
from oracle import pyref
from oracle.pyref import P

pytestmark = pytest.mark.gpu


def tiles_from_streams(synth, sym_stream, col_stream):
    """sym_stream (F,6200) / col_stream (F,3100) uint8 -> tile index per linear cell (what Encoder::encode_next lays out)."""
    f = sym_stream.shape[0]
    sym = torch.from_numpy(sym_stream.astype(np.int64))
    col = torch.from_numpy(col_stream.astype(np.int64))
    sym_cells = torch.stack([sym >> 4, sym & 15], dim=2).reshape(f, modeb.NCELLS)
    col_cells = torch.stack([(col >> 6) & 3, (col >> 4) & 3, (col >> 2) & 3, col & 3], dim=2).reshape(f, modeb.NCELLS)
    out = torch.empty((f, modeb.NCELLS), dtype=torch.int64)
    out[:, synth.stream_cell] = col_cells * 16 + sym_cells
    return out


def test_rs_blocks_with_known_error_patterns(hip_decoder, synth, oracle):
    g = np.random.default_rng(99)
    nframes = 12
    msgs = g.integers(0, 256, (nframes * 60, 125), dtype=np.uint8)
    blocks = framegen.rs_encode(torch.from_numpy(msgs)).numpy().copy()
    for b in range(blocks.shape[0]):
        ne = int(g.integers(0, 23))
        mode = b % 4
        if mode == 0:
            pos = g.choice(155, ne, replace=False)
        elif mode == 1:   # burst at the front
            pos = np.arange(ne)
        elif mode == 2:   # burst over the parity tail, including the very last byte (error location 0)
            pos = 154 - np.arange(ne)
        else:             # message/parity boundary
            pos = (118 + np.arange(ne)) % 155
        blocks[b, pos] ^= g.integers(1, 256, ne, dtype=np.uint8)
    blocks = blocks.reshape(nframes, 60, 155)
    sym_stream = blocks[:, :40].reshape(nframes, 6200)
    col_stream = blocks[:, 40:].reshape(nframes, 3100)
    frames = synth.render(tiles_from_streams(synth, sym_stream, col_stream)).numpy()

    hip_decoder.reset_ccm()
    # colour correction off: the colour stream must reach RS exactly as rendered, whatever the (garbage) headers say
    total, chunks, masks = hip_decoder.decode_batch(frames, color_correction=0)
    rs_ok = hip_decoder.tap(D.TAP_RS_OK, nframes)
    sym = hip_decoder.tap(D.TAP_SYMBOLS, nframes)
    col = hip_decoder.tap(D.TAP_COLORS, nframes)
    want_tiles = tiles_from_streams(synth, sym_stream, col_stream).numpy()
    assert (col.astype(np.int64) * 16 + sym == want_tiles).all(), "cells must reach RS exactly as rendered"

    ccm = pyref.CoCcm()
    for f in range(nframes):
        r, want_chunks, want_mask, ccm = pyref.oracle_decode(frames[f], 0, 0, ccm)
        for b in range(60):
            out = np.zeros(125, np.uint8)
            rr = oracle.co_rs_decode(P(np.ascontiguousarray(blocks[f, b])), 155, 30, P(out))
            assert bool(rs_ok[f, b]) == (rr > 0), f"frame {f} block {b}: ok flag {rs_ok[f, b]} vs libcorrect {rr}"
        assert masks[f] == want_mask, f"frame {f}: mask {masks[f]:#x} vs {want_mask:#x}"
        assert (chunks[f] == want_chunks).all(), f"frame {f}: chunk bytes differ"
