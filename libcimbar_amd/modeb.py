"""Mode-B ("Conf8x8") constants and index maps, host side.

Every value restates /root/reference/src/lib/cimb_translator/GridConf.h:121-142 and Config.h:101-165; the maps restate
CellPositions.cpp:5-51 and Interleave.h:8-36. tests/test_modeb_tables.py checks them against the reference's golden values
(InterleaveTest.cpp, FloodDecodePositionsTest.cpp) and, where oracle/_ref is present, against the reference build.
"""
import numpy as np

IMG = 1024
CELL = 8
PITCH = 9
OFFSET = 8
DIM = 112
MARKER = 6                      # lrint(54 / 9), GridConf.h:32-40
TOP_W = DIM - 2 * MARKER        # 100
TOP_CELLS = TOP_W * MARKER      # 600
MID_CELLS = DIM * (DIM - 2 * MARKER)
NCELLS = 12400
RS_BLOCK = 155
RS_PARITY = 30
RS_DATA = 125
SYM_BLOCKS = 40
COL_BLOCKS = 20
CHUNK = 625
CHUNKS_PER_FRAME = 12
FRAME_BYTES = CHUNK * CHUNKS_PER_FRAME   # 7500
FRAME_RGB_BYTES = IMG * IMG * 3

# Common.cpp:21-31 getColor4 (colour_mode 1)
PALETTE = np.array([[0, 255, 0], [0, 255, 255], [255, 255, 0], [255, 0, 255]], dtype=np.uint8)

# CimbDecoder.cpp:87-99: average_hash of the 16 embedded 8x8 tiles; bit 63 = top-left pixel, 1 = foreground
TILE_HASHES = np.array([
    0xfffefcf8f0e0c080, 0x80c0e0f0f8fcfeff, 0xff7f3f1f0f070301, 0x0103070f1f3f7fff,
    0x181818ffff181818, 0x66e7e70000e7e766, 0x3c7ee7c3c3e77e3c, 0x18183c3c7e7effff,
    0xc0f0fcfffffcf0c0, 0xfffcf00000f0fcff, 0xff3f0f00000f3fff, 0xe7e7e7e7c3c38181,
    0x8181c3c3e7e7e7e7, 0x0000c3e77e3c1800, 0x0c1c387070381c0c, 0x1e1e38381c1c7878], dtype=np.uint64)


def cell_positions():
    """(NCELLS, 2) int32 top-left pixel (x, y) of every cell in linear order (CellPositions.cpp:5-51)."""
    xy = np.empty((NCELLS, 2), dtype=np.int32)
    i = np.arange(TOP_CELLS)
    xy[:TOP_CELLS, 0] = (i % TOP_W) * PITCH + PITCH * MARKER + OFFSET
    xy[:TOP_CELLS, 1] = (i // TOP_W) * PITCH + OFFSET
    j = np.arange(MID_CELLS)
    xy[TOP_CELLS:TOP_CELLS + MID_CELLS, 0] = (j % DIM) * PITCH + OFFSET
    xy[TOP_CELLS:TOP_CELLS + MID_CELLS, 1] = (j // DIM) * PITCH + MARKER * PITCH + OFFSET
    xy[TOP_CELLS + MID_CELLS:, 0] = (i % TOP_W) * PITCH + PITCH * MARKER + OFFSET
    xy[TOP_CELLS + MID_CELLS:, 1] = (i // TOP_W) * PITCH + (DIM - MARKER) * PITCH + OFFSET
    return xy


def cell_rowcol():
    """(NCELLS, 2) int32 (row, col) of every cell on the 112x112 grid."""
    xy = cell_positions()
    return np.stack([(xy[:, 1] - OFFSET) // PITCH, (xy[:, 0] - OFFSET) // PITCH], axis=1).astype(np.int32)


def interleave_indices(size=NCELLS, num_chunks=RS_BLOCK, partitions=2):
    """stream index -> linear cell index (Interleave.h:8-24)."""
    if num_chunks == 0:
        return np.arange(size, dtype=np.uint32)
    part_size = size // partitions
    out = []
    for part in range(0, size, part_size):
        for chunk in range(num_chunks):
            out.append(np.arange(chunk, part_size, num_chunks, dtype=np.uint32) + part)
    return np.concatenate(out)


def interleave_reverse(size=NCELLS, num_chunks=RS_BLOCK, partitions=2):
    """linear cell index -> stream index (Interleave.h:26-36)."""
    idx = interleave_indices(size, num_chunks, partitions)
    inv = np.zeros(len(idx), dtype=np.uint32)
    inv[idx] = np.arange(len(idx), dtype=np.uint32)
    return inv


def tile_masks():
    """(16, 8, 8) bool foreground masks of the symbol tiles, from TILE_HASHES (MSB = top-left)."""
    bits = ((TILE_HASHES[:, None] >> np.arange(63, -1, -1, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    return bits.reshape(16, 8, 8)


def make_header(encode_id, file_size, block_id):
    """6-byte fountain chunk header (FountainMetadata.h:18-24 to_uint8_arr)."""
    return bytes([(encode_id & 0x7F) | ((file_size >> 17) & 0x80), (file_size >> 16) & 0xFF, (file_size >> 8) & 0xFF,
                  file_size & 0xFF, (block_id >> 8) & 0xFF, block_id & 0xFF])
