"""Synthetic mode-B frame generator (the encode half of the reference, restated as batched tensor ops).

Used to manufacture inputs for tests and bench.py -- it is NOT part of the decode path. Pure torch, device-agnostic:
on the GPU box it renders a 1024-frame batch in HBM in well under a second; on CPU it is used for small test batches.

What it restates (paths relative to /root/reference/src):
  * lib/encoder/Encoder.h:69-129 encode_next: 40 RS blocks fill the 4-bit symbol lane, 20 fill the 2-bit colour lane
  * third_party_lib/libcorrect/src/reed-solomon/encode.c:3-34: systematic RS(155,125), GF(2^8) poly 0x187, roots alpha^1..alpha^30
  * lib/cimb_translator/CimbWriter.cpp:84-95 + CellPositions.cpp:52-58: stream cell k is pasted at cell interleave_indices[k]
  * lib/cimb_translator/Common.cpp:150-171 getTile: foreground pixels take the palette colour, background black (dark mode)
The background/anchor/guide template is a fixture rendered once by the reference build (libcimbar_amd/data/modeb_template.npz,
made by oracle/make_assets.py). tests/test_framegen_vs_ref.py checks the output byte-for-byte against Encoder::encode_next.
"""
import os

import numpy as np
import torch

from . import geometry, modeb

_HERE = os.path.dirname(os.path.abspath(__file__))
TEMPLATE_PATH = os.path.join(_HERE, "data", "modeb_template.npz")   # mode B; other modes: geometry.Geometry.TEMPLATE


def _gf_tables():
    exp = np.zeros(512, dtype=np.int64)
    log = np.zeros(256, dtype=np.int64)
    element = 1
    exp[0] = 1
    for i in range(1, 512):
        element *= 2
        if element > 255:
            element ^= 0x187
        exp[i] = element
        if i < 256:
            log[element] = i
    return exp, log


_EXP, _LOG = _gf_tables()


def _generator(parity=modeb.RS_PARITY):
    """g(x) = prod_{i=1..parity} (x + alpha^i), coefficients low -> high (reed-solomon.c:5-12 + polynomial.c:240+)."""
    def mul(a, b):
        return 0 if a == 0 or b == 0 else int(_EXP[_LOG[a] + _LOG[b]])
    gen = [1] + [0] * parity
    for i in range(parity):
        root = int(_EXP[(i + 1) % 255])
        for j in range(i + 1, 0, -1):
            gen[j] = gen[j - 1] ^ mul(gen[j], root)
        gen[0] = mul(gen[0], root)
    return gen


def rs_encode(msgs, parity=modeb.RS_PARITY):
    """msgs: (N, k) uint8 tensor -> (N, k + parity) uint8 (message followed by the parity bytes): (N,125) -> (N,155) in mode B."""
    dev = msgs.device
    exp = torch.from_numpy(_EXP).to(dev)
    log = torch.from_numpy(_LOG).to(dev)
    gen = _generator(parity)
    gen_log = torch.tensor([int(_LOG[g]) if g else -1 for g in gen[:parity]], device=dev)
    n = msgs.shape[0]
    rem = torch.zeros((n, parity), dtype=torch.int64, device=dev)
    m = msgs.to(torch.int64)
    for i in range(msgs.shape[1]):
        fb = m[:, i] ^ rem[:, parity - 1]
        prod = exp[(log[fb][:, None] + gen_log.clamp(min=0)[None, :])]
        prod = torch.where((fb[:, None] == 0) | (gen_log[None, :] < 0), torch.zeros_like(prod), prod)
        rem = torch.cat([prod[:, :1], rem[:, :-1] ^ prod[:, 1:]], dim=1)
    return torch.cat([m, rem.flip(1)], dim=1).to(torch.uint8)


class FrameSynth:
    """Renders batches of clean frames of one mode from frame payloads (12 fountain chunks per frame: 12 x 625 bytes in mode B,
    12 x 429 in mode Bm)."""

    def __init__(self, device="cpu", mode=68):
        self.device = torch.device(device)
        self.geo = g = geometry.for_mode(mode)
        path = os.path.join(_HERE, "data", g.TEMPLATE)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `python oracle/make_assets.py` where /root/reference exists")
        z = np.load(path)
        template = np.zeros(g.FRAME_RGB_BYTES, dtype=np.uint8)
        template[z["idx"]] = z["val"]
        self.template = torch.from_numpy(template.reshape(g.FRAME_SHAPE)).to(self.device)
        masks = torch.from_numpy(modeb.tile_masks()).to(self.device)                      # (16,8,8) bool
        pal = torch.from_numpy(g.PALETTE).to(self.device)                                 # (4,3)
        # tile[colour*16 + symbol] (64 | 128,8,8,3): Common.cpp:150-171
        self.tiles = (masks[None, :, :, :, None] * pal[:, None, None, None, :]).reshape(16 * pal.shape[0], 8, 8, 3).to(torch.uint8)
        self.stream_cell = torch.from_numpy(g.interleave_indices().astype(np.int64)).to(self.device)
        xy = g.cell_positions()
        self.cell_row = torch.from_numpy(((xy[:, 1] - g.OFFSET) // g.PITCH).astype(np.int64)).to(self.device)
        self.cell_col = torch.from_numpy(((xy[:, 0] - g.OFFSET) // g.PITCH).astype(np.int64)).to(self.device)

    def cell_tiles(self, payload):
        """payload (F,FRAME_BYTES) uint8 -> tile index (colour*16+symbol) per linear cell, (F,NCELLS) int64."""
        g = self.geo
        payload = payload.to(self.device)
        f = payload.shape[0]
        blocks = rs_encode(payload.reshape(f * g.BLOCKS, g.RS_DATA), g.RS_PARITY).reshape(f, g.BLOCKS * g.RS_BLOCK).to(torch.int64)
        if g.LEGACY:
            # Encoder::encode_next_coupled (Encoder.h:131-163): the RS-encoded stream read 6 bits at a time, stream cell s = colour(2) | symbol(4)
            cb = 4 + g.COLOR_BITS                                                            # 6 | 7 bits per cell
            bits = ((blocks[:, :, None] >> torch.arange(7, -1, -1, device=self.device)) & 1).reshape(f, g.NCELLS, cb)
            w6 = 2 ** torch.arange(cb - 1, -1, -1, device=self.device)
            tiles_stream = (bits * w6).sum(dim=2)                                           # = colour * 16 + symbol
            out = torch.empty_like(tiles_stream)
            out[:, self.stream_cell] = tiles_stream
            return out
        sym = blocks[:, :g.SYM_BLOCKS * g.RS_BLOCK]                                        # (F,6200)
        col = blocks[:, g.SYM_BLOCKS * g.RS_BLOCK:]                                        # (F,3100)
        sym_cells = torch.stack([sym >> 4, sym & 15], dim=2).reshape(f, g.NCELLS)           # stream order, high nibble first
        col_cells = torch.stack([(col >> 6) & 3, (col >> 4) & 3, (col >> 2) & 3, col & 3], dim=2).reshape(f, g.NCELLS)
        tiles_stream = col_cells * 16 + sym_cells
        out = torch.empty_like(tiles_stream)
        out[:, self.stream_cell] = tiles_stream                                             # stream cell k sits at cell stream_cell[k]
        return out

    def render(self, tile_idx, out=None):
        """tile_idx (F,NCELLS) int64 -> frames (F,IMG_H,IMG_W,3) uint8."""
        g = self.geo
        f = tile_idx.shape[0]
        if out is None:
            out = torch.empty((f, *g.FRAME_SHAPE), dtype=torch.uint8, device=self.device)
        out[:] = self.template
        # a (DIM_Y*9 x DIM_X*9) canvas view starting at the grid origin: cell (r,c) is canvas[r, :8, c, :8]
        region = out[:, g.OFFSET:g.OFFSET + g.DIM_Y * g.PITCH, g.OFFSET:g.OFFSET + g.DIM_X * g.PITCH, :]
        grid = torch.zeros((f, g.DIM_Y, g.DIM_X), dtype=torch.int64, device=self.device)
        present = torch.zeros((g.DIM_Y, g.DIM_X), dtype=torch.bool, device=self.device)
        grid[:, self.cell_row, self.cell_col] = tile_idx
        present[self.cell_row, self.cell_col] = True
        cells = self.tiles[grid]                                                            # (F,112,112,8,8,3)
        view = region.reshape(f, g.DIM_Y, g.PITCH, g.DIM_X, g.PITCH, 3)
        old = view[:, :, :8, :, :8, :].permute(0, 1, 3, 2, 4, 5)                             # (F,112,112,8,8,3)
        merged = torch.where(present[None, :, :, None, None, None], cells, old)
        view[:, :, :8, :, :8, :] = merged.permute(0, 1, 3, 2, 4, 5)
        return out

    def frames_from_payload(self, payload, out=None):
        return self.render(self.cell_tiles(payload), out=out)


def synth_payload(n_frames, seed=1234, encode_id=1, file_size=None, first_block=0, device="cpu", mode=68):
    """Deterministic stand-in for a fountain stream: n_frames*12 chunks of [6-byte header | 619 (mode Bm: 423) random bytes].
    Headers follow FountainMetadata.h:18-24 with consecutive block ids, so CimbReader::update_metadata's prediction
    (CimbReader.cpp:269-280) holds and the colour-correction path is exercised the same way as on a real stream."""
    geo = geometry.for_mode(mode)
    per = geo.CHUNKS_PER_FRAME
    if file_size is None:
        file_size = n_frames * per * (geo.CHUNK - 6)
    g = np.random.default_rng(seed)
    chunks = g.integers(0, 256, size=(n_frames * per, geo.CHUNK), dtype=np.uint8)
    ids = (np.arange(n_frames * per) + first_block) & 0xFFFF
    chunks[:, 0] = (encode_id & 0x7F) | ((file_size >> 17) & 0x80)
    chunks[:, 1] = (file_size >> 16) & 0xFF
    chunks[:, 2] = (file_size >> 8) & 0xFF
    chunks[:, 3] = file_size & 0xFF
    chunks[:, 4] = (ids >> 8) & 0xFF
    chunks[:, 5] = ids & 0xFF
    return torch.from_numpy(chunks.reshape(n_frames, geo.FRAME_BYTES)).to(device)


def inject_cell_errors(tile_idx, n_errors=99, seed=5678):
    """BASELINE config 3: per frame replace `n_errors` distinct cells by a DIFFERENT valid tile (new symbol uniform over the
    other 15, new colour uniform over 4). tile_idx (F,12400) int64 -> new tensor."""
    out = tile_idx.clone()
    f = out.shape[0]
    for k in range(f):
        g = np.random.default_rng(seed + k)
        cells = g.choice(out.shape[1], size=n_errors, replace=False)
        old = out[k, torch.from_numpy(cells).to(out.device)].cpu().numpy()
        new_sym = (old % 16 + g.integers(1, 16, size=n_errors)) % 16
        new_col = g.integers(0, 4, size=n_errors)          # (a valid colour in every mode)
        out[k, torch.from_numpy(cells).to(out.device)] = torch.from_numpy(new_col * 16 + new_sym).to(out.device)
    return out
