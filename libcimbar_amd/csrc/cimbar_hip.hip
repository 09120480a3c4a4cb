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
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cimbar_hip.h"

// One copy of the geometry-dependent code per supported mode (Config.h:19-44), then the mode-independent C ABI on top.
#define CIMBAR_MODE 68
#define CIMBAR_NS m68
#include "mode.hip.inc"
#undef CIMBAR_MODE
#undef CIMBAR_NS
#define CIMBAR_MODE 67
#define CIMBAR_NS m67
#include "mode.hip.inc"
#undef CIMBAR_MODE
#undef CIMBAR_NS
#define CIMBAR_MODE 66
#define CIMBAR_NS m66
#include "mode.hip.inc"
#undef CIMBAR_MODE
#undef CIMBAR_NS
#define CIMBAR_MODE 4
#define CIMBAR_NS m4
#include "mode.hip.inc"
#undef CIMBAR_MODE
#undef CIMBAR_NS
#define CIMBAR_MODE 8
#define CIMBAR_NS m8
#include "mode.hip.inc"
#undef CIMBAR_MODE
#undef CIMBAR_NS

#include "api.hip.inc"
#include "comm.hip.inc"
#include "png.hip.inc"
