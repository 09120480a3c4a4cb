// cimbar_hip.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI for libcimbar's mode-B frame-decode path.
//
// One deskewed 1024x1024 RGB8 frame -> <=12 fountain chunks of 625 bytes, bit-exact with the reference's
// Decoder::decode_fountain (src/lib/encoder/Decoder.h:171-189). See include/cimbar_hip.h for the boundary and
// DESIGN.md for the data layout / roofline notes. Paths in comments are relative to /root/reference/src.
//
// Pipeline (a batch of F frames per launch; nothing here is GEMM-shaped, so no MFMA). One call = K1, then the chain of short
// kernels, on the caller's stream (large batches: the chain as two half-batches on two streams); the pipelined entry point
// alternates whole batches between two context-owned streams. See enqueue() and DESIGN.md "Launch structure".
//   K1 k_threshold      RGB -> gray -> (sharpen) -> 5x5|7x7 box-mean threshold -> bitplane   [HBM-read bound]
//   K2 k_symbols        every cell at drift (0,0): 10x10 bit window -> 5|9 shifted 8x8 hashes -> popcount match;
//                       flags the frame if any cell prefers a shifted window (order then matters -> K2b)
//   K2c k_flood_wave    batch-parallel flood for frames whose result provably does not depend on the reference's tie order
//   K2b k_flood         exact emulation of the reference's priority-flood order + drift, one wavefront per flagged frame,
//                       heap and per-cell state in LDS
//   K3 k_rs             de-interleave + RS(155,125) decode, one block per wavefront (symbols: 40 blocks)
//   K4 k_frame_mid      aligned_stream bookkeeping for the symbol chunks, fountain-header prediction, CCM derivation
//   K5 k_colors         6x6 cell mean -> CCM -> palette classifier
//   K3 k_rs             (colours: 20 blocks)
//   K7 k_frame_end      aligned_stream bookkeeping for the colour chunks, chunk mask, zero dropped slots, CCM carry-out
//   E1 k_rs_encode / E2 k_render   the encode half (Encoder::encode_next): RS encode + tile render
//   X1-X4 k_scan_* / k_warp        the stage in front (Scanner's image preparation, Deskewer's perspective warp)
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cimbar_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ mode-B constants
// lib/cimb_translator/GridConf.h:121-142 (Conf8x8); Config.h:101-165
constexpr int IMG = 1024;
constexpr int PITCH = 9, OFFSET = 8, DIM = 112, MARKER = 6;
constexpr int TOP_W = DIM - 2 * MARKER;               // 100
constexpr int TOP_CELLS = TOP_W * MARKER;             // 600
constexpr int MID_CELLS = DIM * (DIM - 2 * MARKER);   // 11200
constexpr int NCELLS = 12400;
constexpr int RS_BLOCK = 155, RS_PARITY = 30, RS_DATA = 125;
constexpr int SYM_BLOCKS = 40, COL_BLOCKS = 20, ALL_BLOCKS = 60;
constexpr int CHUNK = 625, CHUNKS = 12, FRAME_BYTES = CHUNK * CHUNKS;
constexpr int PLANE_WORDS = IMG * IMG / 32;           // 32768 u32 per frame
constexpr size_t FRAME_RGB = (size_t)IMG * IMG * 3;
constexpr int ANCHOR = 30;
constexpr int HEAP_CAP = 12 * NCELLS + 64;            // <= 12 offers per decoded cell + 8 seeds

// the 16 tile hashes and the exact-match slot table of k_symbols: computed by cimbar_hip_create from the tile bitmaps the way
// CimbDecoder's constructor does (CimbDecoder.cpp:58-66,87-99 -> Common.cpp:150-171 getTile -> average_hash.h:19-39), see
// build_tile_hashes() in host.hip.inc
__constant__ uint64_t c_tile[16];

// Common.cpp:21-31 getColor4 (colour_mode 1)
__constant__ int c_palette[4][3] = {{0, 255, 0}, {0, 255, 255}, {255, 255, 0}, {255, 0, 255}};

// GF(2^8), primitive poly 0x187 (libcorrect field.h:26-62): exp[512], log[256]
__constant__ uint8_t c_gf_exp[512];
__constant__ uint8_t c_gf_log[256];

struct Tables {
	ushort2* cell_xy;        // [NCELLS] top-left pixel of each cell (CellPositions.cpp:5-51)
	uint16_t* stream_cell;   // [NCELLS] stream index -> linear cell index (Interleave.h:8-24)
	uint16_t* cell_grid;     // [NCELLS] grid slot (row * 112 + col) of each cell: where K1 left its 6x6 colour mean
	uint16_t* ccm_grid;      // [96] grid slot (row * 112 + col) of the cells whose colour the fountain header predicts: colour-stream
	                         // cells 3100*c + t, t < 24 (CimbReader.cpp:188-227)
	int16_t* cand;           // [NCELLS][12] the cells FloodDecodePositions::update may offer to, in its order: right, left, bottom,
	                         // top (AdjacentCellFinder.cpp:16-105), then the 4 horizontal and 4 vertical "horizon" cells
	                         // (FloodDecodePositions.cpp:93-129); -1 = none
};

#include "k1_threshold.hip.inc"
#include "k2_symbols.hip.inc"
#include "k2b_flood.hip.inc"
#include "k2c_floodwave.hip.inc"
#include "k3_rs.hip.inc"
#include "k4_frame.hip.inc"
#include "encode.hip.inc"
#include "extract.hip.inc"
#include "scan.hip.inc"

}  // namespace

#include "host.hip.inc"
#include "comm.hip.inc"
