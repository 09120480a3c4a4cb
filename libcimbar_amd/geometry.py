"""The grid of a cimbar mode, host side: what the reference keeps in cimbar::conf (GridConf.h:9-72) and reads through Config:: getters
(Config.h:52-165). Modes built into the HIP library: 68 ("B", Conf8x8, GridConf.h:121-142), 67 ("Bm", Conf8x8_mini, GridConf.h:168-189) and
66 ("Bu", Conf8x8_micro, GridConf.h:144-166), and the legacy 4-colour mode 4 ("4C", Config.h:24-29: Conf8x8 with the coupled decode).
`modeb` is the mode-68 instance spelled out as module constants; tests/test_modeb_tables.py checks the two agree.
"""
from dataclasses import dataclass

import numpy as np

_CONF = {
    # mode: (image_size_x, image_size_y, cell_offset, cells_per_col_x, cells_per_col_y, ecc_block_size, ecc_bytes, fountain_chunks_scalar)
    68: (1024, 1024, 8, 112, 112, 155, 30, 2),
    67: (1024, 720, 9, 112, 78, 179, 36, 2),
    66: (736, 637, 9, 80, 69, 168, 33, 1),
    4: (1024, 1024, 8, 112, 112, 155, 30, -10),     # Config.h:24-29: Conf8x8 + legacy_mode, fountain_chunks_scalar -10 = ten chunks per frame
    8: (1024, 1024, 8, 112, 112, 155, 30, -10),     # Config.h:30-35: the same with 3 colour bits (8 colours)
}


@dataclass(frozen=True)
class Geometry:
    MODE: int
    IMG_W: int
    IMG_H: int
    OFFSET: int
    DIM_X: int
    DIM_Y: int
    RS_BLOCK: int
    RS_PARITY: int
    CELL: int = 8
    PITCH: int = 9
    MARKER: int = 6                # lrint(54 / 9), GridConf.h:32-40
    SYMBOL_BITS: int = 4
    COLOR_BITS: int = 2
    CHUNKS_PER_FRAME: int = 12     # fountain_chunks_per_frame = bits_per_cell * fountain_chunks_scalar (2; micro: 1), GridConf.h:54-61
    LEGACY: bool = False           # legacy_mode (Config.h:24-35): symbol and colour bits coupled in ONE Reed-Solomon stream, colour_mode 0 palette

    # ---- derived (GridConf.h:42-71)
    @property
    def TOP_W(self): return self.DIM_X - 2 * self.MARKER
    @property
    def TOP_CELLS(self): return self.TOP_W * self.MARKER
    @property
    def MID_CELLS(self): return self.DIM_X * (self.DIM_Y - 2 * self.MARKER)
    @property
    def NCELLS(self): return self.DIM_X * self.DIM_Y - 4 * self.MARKER * self.MARKER
    @property
    def RS_DATA(self): return self.RS_BLOCK - self.RS_PARITY
    @property
    def SYM_BLOCKS(self):
        return self.NCELLS * (self.SYMBOL_BITS + self.COLOR_BITS) // 8 // self.RS_BLOCK if self.LEGACY else self.NCELLS * self.SYMBOL_BITS // 8 // self.RS_BLOCK
    @property
    def COL_BLOCKS(self): return 0 if self.LEGACY else self.NCELLS * self.COLOR_BITS // 8 // self.RS_BLOCK
    @property
    def BLOCKS(self): return self.SYM_BLOCKS + self.COL_BLOCKS
    @property
    def CHUNK(self):
        cap = self.NCELLS * (self.SYMBOL_BITS + self.COLOR_BITS) // 8
        return cap * self.RS_DATA // self.RS_BLOCK // self.CHUNKS_PER_FRAME
    @property
    def FRAME_BYTES(self): return self.CHUNK * self.CHUNKS_PER_FRAME
    @property
    def FRAME_RGB_BYTES(self): return self.IMG_W * self.IMG_H * 3
    @property
    def FRAME_SHAPE(self): return (self.IMG_H, self.IMG_W, 3)
    @property
    def TEMPLATE(self): return {68: "modeb_template.npz", 67: "modebm_template.npz", 66: "modebu_template.npz", 4: "modeb_template.npz", 8: "modeb_template.npz"}[self.MODE]
    @property
    def PALETTE(self):
        """Common.cpp:21-84: getColor4 (colour_mode 1) / getColor4_old, getColor8_old (colour_mode 0, the legacy modes)"""
        if self.LEGACY and self.COLOR_BITS == 3:
            return np.array([[0, 255, 255], [127, 127, 255], [255, 0, 255], [255, 65, 65], [255, 159, 0], [255, 255, 0], [255, 255, 255], [0, 255, 0]], dtype=np.uint8)
        return np.array([[0, 255, 255], [255, 255, 0], [255, 0, 255], [0, 255, 0]] if self.LEGACY else [[0, 255, 0], [0, 255, 255], [255, 255, 0], [255, 0, 255]], dtype=np.uint8)
    @property
    def FULL_MASK(self): return (1 << self.CHUNKS_PER_FRAME) - 1

    def cell_positions(self):
        """(NCELLS, 2) int32 top-left pixel (x, y) of every cell in linear order (CellPositions.cpp:5-51)."""
        xy = np.empty((self.NCELLS, 2), dtype=np.int32)
        T, M = self.TOP_CELLS, self.MID_CELLS
        i = np.arange(T)
        xy[:T, 0] = (i % self.TOP_W) * self.PITCH + self.PITCH * self.MARKER + self.OFFSET
        xy[:T, 1] = (i // self.TOP_W) * self.PITCH + self.OFFSET
        j = np.arange(M)
        xy[T:T + M, 0] = (j % self.DIM_X) * self.PITCH + self.OFFSET
        xy[T:T + M, 1] = (j // self.DIM_X) * self.PITCH + self.MARKER * self.PITCH + self.OFFSET
        xy[T + M:, 0] = (i % self.TOP_W) * self.PITCH + self.PITCH * self.MARKER + self.OFFSET
        xy[T + M:, 1] = (i // self.TOP_W) * self.PITCH + (self.DIM_Y - self.MARKER) * self.PITCH + self.OFFSET
        return xy

    def interleave_indices(self):
        """stream index -> linear cell index (Interleave.h:8-24 with ecc_block_size blocks, 2 partitions; Config.h:157-165)."""
        part = self.NCELLS // 2
        out = []
        for p in range(0, self.NCELLS, part):
            for c in range(self.RS_BLOCK):
                out.append(np.arange(c, part, self.RS_BLOCK, dtype=np.uint32) + p)
        return np.concatenate(out)


def for_mode(mode=68):
    """Config::temp_conf(mode_val) for the modes the HIP library is built for (0 = the default, mode B)."""
    mode = 68 if mode in (0, None) else int(mode)
    if mode not in _CONF:
        raise ValueError(f"cimbar mode {mode} is not built (supported: 68 'B', 67 'Bm', 66 'Bu', 4 '4C', 8 '8C')")
    w, h, off, dx, dy, blk, par, scalar = _CONF[mode]
    return Geometry(mode, w, h, off, dx, dy, blk, par, COLOR_BITS=3 if mode == 8 else 2, CHUNKS_PER_FRAME=6 * scalar if scalar > 0 else -scalar,
                    LEGACY=mode in (4, 8))
