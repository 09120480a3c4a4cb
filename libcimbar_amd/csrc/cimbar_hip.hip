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
//   K2b k_flood         exact emulation of the reference's priority-flood order + drift, one wavefront per flagged frame,
//                       heap and per-cell state in LDS
//   K3 k_rs             de-interleave + RS(155,125) decode, one block per wavefront (symbols: 40 blocks)
//   K4 k_frame_mid      aligned_stream bookkeeping for the symbol chunks, fountain-header prediction, CCM derivation
//   K5 k_colors         6x6 cell mean -> CCM -> palette classifier
//   K3 k_rs             (colours: 20 blocks)
//   K7 k_frame_end      aligned_stream bookkeeping for the colour chunks, chunk mask, zero dropped slots, CCM carry-out
//   E1 k_rs_encode / E2 k_render   the encode half (Encoder::encode_next): RS encode + tile render
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cfloat>
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

// the 16 tile hashes (CimbDecoder.cpp:87-99 computes them from bitmaps.h; checked against the reference build in tests)
__constant__ uint64_t c_tile[16] = {
	0xfffefcf8f0e0c080ULL, 0x80c0e0f0f8fcfeffULL, 0xff7f3f1f0f070301ULL, 0x0103070f1f3f7fffULL,
	0x181818ffff181818ULL, 0x66e7e70000e7e766ULL, 0x3c7ee7c3c3e77e3cULL, 0x18183c3c7e7effffULL,
	0xc0f0fcfffffcf0c0ULL, 0xfffcf00000f0fcffULL, 0xff3f0f00000f3fffULL, 0xe7e7e7e7c3c38181ULL,
	0x8181c3c3e7e7e7e7ULL, 0x0000c3e77e3c1800ULL, 0x0c1c387070381c0cULL, 0x1e1e38381c1c7878ULL};

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

// ------------------------------------------------------------------------------------------------ K1 threshold+pack
// CimbReader.cpp:30-46 preprocessSymbolGrid + bitmatrix.h:14-46, fused with the colour pass's pixel reads.
// One wavefront owns a full-width strip of pixel rows: lane l holds columns [16l, 16l+16) and the wave walks down the
// strip, so every RGB byte is loaded from HBM exactly once (3 x 16 B per lane per row, next row prefetched while the
// current one is processed); the 5x5|7x7 box sums live in registers.
//   gray  = (R*9798 + G*19235 + B*3735 + 2^14) >> 15                                 [assumed-OpenCV]
//   bit   = gray > round(box_sum / n)   <=>   n*gray > box_sum + n/2   (n = 25 | 49), BORDER_REPLICATE
// Output 1: plane[frame][row][32 words], pixel x -> word x/32, bit 31-(x%32).
// Output 2: cellmean[frame][112*112] = r | g<<8 | b<<16, the inner-6x6 mean (Cell.h:30-62: uint16 sums / 36) of every
//           cell at its UNDRIFTED grid position -- what read_color / init_ccm sample when no drift is in play
//           (CimbReader.cpp:133-137,216-217). Strips are aligned to the cell grid (56 strips x 2 cell rows), so a cell's
//           six inner rows always belong to one wave: per-byte column sums accumulate in registers over those rows and
//           are regrouped into cells through a 6 KiB LDS transpose once per cell row.
// Strip height is a trade: tall strips re-read few halo rows (2*RAD per strip) but 1024 frames x 16 strips = 5.33 "rounds" of
// the 3072 resident waves leave the chip partly idle while the last round drains; short strips drain evenly but re-read more.
// Measured on MI355X, 1024-frame batches: 16 strips 0.73 ms, 28: 0.73, 56: 0.68, 112: 0.80. A persistent-wave work queue mixing
// tall and short units was tried and lost to the plain grid (0.79 ms).
constexpr int K1_STRIPS = 56, K1_CELLROWS = DIM / K1_STRIPS;   // 2 cell rows = 18 pixel rows per strip (+8 px margin at both ends of the frame)
static_assert(DIM % K1_STRIPS == 0 && K1_STRIPS % 4 == 0, "strips must tile the cell grid and the 4-wave workgroups");
constexpr int GRID_CELLS = DIM * DIM;

__device__ __forceinline__ void load_row48(const uint8_t* __restrict__ row, int lane, uint32_t d[12])
{
	const uint4* p = reinterpret_cast<const uint4*>(row + lane * 48);
	uint4 a = p[0], b = p[1], c = p[2];
	d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
	d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
	d[8] = c.x; d[9] = c.y; d[10] = c.z; d[11] = c.w;
}

__device__ __forceinline__ uint32_t byte_of(const uint32_t d[12], int k) { return (d[k >> 2] >> (8 * (k & 3))) & 0xFFu; }

// 16 gray pixels of this lane, g[p]
__device__ __forceinline__ void gray16(const uint32_t d[12], uint32_t g[16])
{
#pragma unroll
	for (int p = 0; p < 16; ++p)
		g[p] = (byte_of(d, 3 * p) * 9798u + byte_of(d, 3 * p + 1) * 19235u + byte_of(d, 3 * p + 2) * 3735u + 16384u) >> 15;
}

// 16 gray pixels straight from the 12 raw dwords with v_dot4_u32_u8: pixel p starts at byte 3p; the dot product's fourth
// coefficient is 0, so the dword may carry the next pixel's first byte. The 15-bit coefficients are split c = hi*128 + lo
// (9798 = 76*128+70, 19235 = 150*128+35, 3735 = 29*128+23) and the low dot product is doubled, so that
//   T = (hi_dot << 8) + 2*lo_dot + 32768 = 2 * (R*9798 + G*19235 + B*3735 + 16384)   and   gray = T >> 16 = byte 2 of T.
// Leaving the result byte-aligned lets one v_perm_b32 pack two pixels into a u16 pair.
__device__ __forceinline__ void gray16_T(const uint32_t d[12], uint32_t T[16])
{
	constexpr uint32_t LO2 = 140u | (70u << 8) | (46u << 16), HI = 76u | (150u << 8) | (29u << 16);
#pragma unroll
	for (int p = 0; p < 16; ++p) {
		const int w = (3 * p) >> 2, sh = (3 * p) & 3;
		uint32_t v;
		if (sh == 0) v = d[w];
		else if (w == 11) v = d[11] >> (8 * sh);
		else v = __builtin_amdgcn_alignbyte(d[w + 1], d[w], sh);
		const uint32_t lo = __builtin_amdgcn_udot4(v, LO2, 32768u, false);
		const uint32_t hi = __builtin_amdgcn_udot4(v, HI, 0u, false);
		T[p] = (hi << 8) + lo;
	}
}
__device__ __forceinline__ void gray16_dot(const uint32_t d[12], uint32_t g[16])
{
	uint32_t T[16];
	gray16_T(d, T);
#pragma unroll
	for (int p = 0; p < 16; ++p) g[p] = T[p] >> 16;
}
// (pixel i | pixel i+8 << 16) from the byte-2 results of gray16_T: one v_perm_b32 per pair
__device__ __forceinline__ void pairs_from_T(const uint32_t T[16], uint32_t G[8])
{
#pragma unroll
	for (int i = 0; i < 8; ++i) G[i] = __builtin_amdgcn_perm(T[i + 8], T[i], 0x0c060c02u);   // bytes: [T_i.b2, 0, T_{i+8}.b2, 0]
}
__device__ __forceinline__ void pairs_from_g(const uint32_t g[16], uint32_t G[8])
{
#pragma unroll
	for (int i = 0; i < 8; ++i) G[i] = g[i] | (g[i + 8] << 16);
}

// lane l <- lane l-1 (lane 0 keeps `edge`), lane l <- lane l+1 (lane 63 keeps `edge`): DPP wave shifts, one VALU op each
__device__ __forceinline__ uint32_t from_left_lane(uint32_t v, uint32_t edge) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t from_right_lane(uint32_t v, uint32_t edge) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false); }

__device__ __forceinline__ const uint8_t* row_ptr(const uint8_t* __restrict__ frame, int r) { return frame + (size_t)r * (IMG * 3); }

// CimbReader.cpp:17-27 sharpen: filter2D with [0 -1 0; -1 4.5 -1; 0 -1 0], BORDER_REFLECT_101, saturate(round-half-even).
// dc = raw bytes of row r (for the colour sums), s = sharpened gray of row r.
__device__ __forceinline__ void sharp_row(const uint8_t* __restrict__ frame, int r, int lane, uint32_t dc[12], uint32_t s[16])
{
	uint32_t dn[12], ds[12], gn[16], gc[16], gs[16];
	int rn = r - 1 < 0 ? 1 : r - 1, rs = r + 1 >= IMG ? IMG - 2 : r + 1;
	load_row48(row_ptr(frame, rn), lane, dn);
	load_row48(row_ptr(frame, r), lane, dc);
	load_row48(row_ptr(frame, rs), lane, ds);
	gray16_dot(dn, gn); gray16_dot(dc, gc); gray16_dot(ds, gs);
	// west / east neighbours with BORDER_REFLECT_101: pixel -1 -> pixel 1, pixel 16 of the last lane -> pixel 14
	const uint32_t west0 = from_left_lane(gc[15], gc[1]), east15 = from_right_lane(gc[0], gc[14]);
#pragma unroll
	for (int p = 0; p < 16; ++p) {
		const uint32_t w = p == 0 ? west0 : gc[p - 1], e = p == 15 ? east15 : gc[p + 1];
		int t = 9 * (int)gc[p] - 2 * (int)(gn[p] + gs[p] + w + e);   // = 2 * (4.5c - n - s - w - e), exact
		int q = t >> 1;
		if (t & 1) q += (q & 1);                                     // x.5 -> nearest even
		s[p] = (uint32_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
	}
}

// The same filter on u16 pairs (pixel i | pixel i+8 << 16) for the streaming path: gn / gc / gs are the gray rows above, at and
// below the output row. Biased by 4096 per half so every intermediate stays a positive 16-bit field:
//   t' = 9c - 2(n+s+w+e) + 4096 in [2056, 6391];  q' = t' >> 1 (+1 if t' odd and q' odd: half -> even; 2048 keeps the parity);
//   result = clamp(q' - 2048, 0, 255).
__device__ __forceinline__ void sharp_pairs(const uint32_t (&gn)[8], const uint32_t (&gc)[8], const uint32_t (&gs)[8], uint32_t (&G)[8])
{
	typedef unsigned short us2 __attribute__((ext_vector_type(2)));
	// BORDER_REFLECT_101 at the row ends: pixel -1 -> pixel 1 (lane 0), pixel 16 of lane 63 -> pixel 14
	const uint32_t W0 = from_left_lane(gc[7] >> 16, gc[1] & 0xFFFFu) | (gc[7] << 16);            // (pixel -1, pixel 7)
	const uint32_t E7 = (gc[0] >> 16) | (from_right_lane(gc[0] & 0xFFFFu, gc[6] >> 16) << 16);   // (pixel 8, pixel 16)
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const uint32_t w = i == 0 ? W0 : gc[i - 1], e = i == 7 ? E7 : gc[i + 1];
		const uint32_t sum = gn[i] + gs[i] + w + e;
		const uint32_t t = gc[i] * 9u + 0x10001000u - 2u * sum;
		uint32_t q = (t >> 1) & 0x7FFF7FFFu;
		q += t & q & 0x00010001u;
		us2 v = __builtin_bit_cast(us2, q);
		const us2 lo = {2048, 2048}, hi = {2048 + 255, 2048 + 255};
		v = __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi);
		G[i] = __builtin_bit_cast(uint32_t, v) - 0x08000800u;
	}
}

// One row of the streaming box-threshold. All per-pixel quantities travel as u16 pairs (pixel i | pixel i+8 << 16), i < 8,
// so a horizontal neighbour is simply the next register and plain 32-bit adds never carry between the halves.
//   ring : the last RING gray rows (RING = 2*RAD+2, even, so that the A/B prefetch buffers keep static roles)
//   C    : per-column sums of the newest 2*RAD+1 rows
// emit: row (y - RAD) has its full window -> 16 result bits for this lane.
template <int RAD, int SLOT>
__device__ __forceinline__ uint32_t box_row(uint32_t (&ring)[2 * RAD + 2][8], uint32_t (&C)[8], const uint32_t G[8], bool emit)
{
	constexpr int RING = 2 * RAD + 2, R = 2 * RAD + 1, N = R * R;
	constexpr int OLD = (SLOT + 1) % RING;            // row y - (2*RAD+1), leaving the window
	constexpr int CTR = (SLOT + RING - RAD) % RING;   // row y - RAD, the one being emitted
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		C[i] += G[i] - ring[OLD][i];
		ring[SLOT][i] = G[i];
	}
	if (!emit) return 0;

	// X[k] = pair register k - RAD, k = 0 .. 8 + 2*RAD - 1: (pixel i, pixel i+8) for i = -RAD .. 7+RAD
	uint32_t X[8 + 2 * RAD];
#pragma unroll
	for (int i = 0; i < 8; ++i) X[RAD + i] = C[i];
#pragma unroll
	for (int k = 1; k <= RAD; ++k) {
		// pixel -k comes from the left lane's pixel 16-k (hi half of its C[8-k]); BORDER_REPLICATE -> own pixel 0 on lane 0
		const uint32_t left = from_left_lane(C[8 - k] >> 16, C[0] & 0xFFFFu);
		X[RAD - k] = left | (C[8 - k] << 16);                       // (pixel -k, pixel 8-k)
		// pixel 15+k comes from the right lane's pixel k-1 (lo half of its C[k-1]); replicate -> own pixel 15 on lane 63
		const uint32_t right = from_right_lane(C[k - 1] & 0xFFFFu, C[7] >> 16);
		X[RAD + 7 + k] = (C[k - 1] >> 16) | (right << 16);          // (pixel 7+k, pixel 15+k)
	}
	uint32_t S = 0;
#pragma unroll
	for (int k = 0; k < R; ++k) S += X[k];
	constexpr uint32_t K = (0x8000u - (uint32_t)(N / 2) - 1u) * 0x00010001u;
	uint32_t M = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		if (i > 0) S += X[i + 2 * RAD] - X[i - 1];
		// N*g > S + N/2  <=>  bit 15 of (N*g + 0x8000 - N/2 - 1 - S), per half; no field under/overflows (|.| <= 12495)
		const uint32_t r = __umul24(ring[CTR][i], (uint32_t)N) + K - S;
		M |= (r >> i) & (0x80008000u >> i);   // v_lshrrev + v_and_or
	}
	return (M & 0xFF00u) | (M >> 24);   // bit 15-p = pixel p
}

template <int RAD, bool PRE>
#ifndef K1_WAVES
#define K1_WAVES 3
#endif
#ifndef K1_PRE_WAVES
#define K1_PRE_WAVES 2
#endif
#ifndef K1_PRE_DEPTH
#define K1_PRE_DEPTH 4
#endif
__global__ __launch_bounds__(256, PRE ? K1_PRE_WAVES : K1_WAVES) void k_threshold(const uint8_t* __restrict__ rgb, uint32_t* __restrict__ plane,
                                                                uint32_t* __restrict__ cellmean, uint32_t* __restrict__ flood_flag, int f0)
{
	constexpr int RING = 2 * RAD + 2;
	__shared__ __attribute__((aligned(16))) uint16_t s_col[4][IMG * 3];   // per-wave column sums of one cell row, by row byte
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps every row index / address base scalar
	const int strip = blockIdx.x * 4 + wave;
	const int f = f0 + blockIdx.y;
	const uint8_t* frame = rgb + (size_t)f * FRAME_RGB;
	uint32_t* out = plane + (size_t)f * PLANE_WORDS;
	uint32_t* cm = cellmean + (size_t)f * GRID_CELLS;
	const int y_begin = strip == 0 ? 0 : OFFSET + strip * K1_CELLROWS * PITCH;
	const int y_end = strip == K1_STRIPS - 1 ? IMG : OFFSET + (strip + 1) * K1_CELLROWS * PITCH;
	const int total = (y_end - y_begin) + 2 * RAD;
	if (strip == 0 && lane == 0) flood_flag[f] = 0;   // per-frame state k_symbols accumulates into

	uint32_t ring[RING][8];
	uint32_t C[8];
	uint32_t acc_e[12], acc_o[12];    // per-byte column sums over a cell's inner rows: even / odd bytes of each dword, u16 x2
#pragma unroll
	for (int k = 0; k < RING; ++k)
#pragma unroll
		for (int i = 0; i < 8; ++i) ring[k][i] = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) C[i] = 0;
#pragma unroll
	for (int k = 0; k < 12; ++k) { acc_e[k] = 0; acc_o[k] = 0; }

	auto clampy = [](int y) { return y < 0 ? 0 : (y >= IMG ? IMG - 1 : y); };   // BORDER_REPLICATE of the thresholded source
	// Even strips walk down, odd strips walk up. A strip needs RAD rows of each neighbour; with alternating directions both
	// owners of a boundary touch it at the same moment (both at their start, or both at their end), so the halo rows are served
	// by L2 instead of being fetched from HBM a second time (measured: 21 % extra read traffic without this).
	const int dir = (strip & 1) ? -1 : 1;
	const int y_first = dir > 0 ? y_begin - RAD : y_end - 1 + RAD;     // first row visited
	const int y_stop = dir > 0 ? y_end - 1 + RAD : y_begin - RAD;      // last row visited
#ifndef K1_DEPTH
#define K1_DEPTH 3
#endif
	constexpr int DEPTH = PRE ? K1_PRE_DEPTH : K1_DEPTH;   // rows in flight per wave; RING % DEPTH == 0 keeps the buffer roles static after unrolling
	static_assert(RING % DEPTH == 0, "prefetch depth must divide the ring size");
	uint32_t buf[DEPTH][12];          // row t lives in buf[t % DEPTH] and is refilled with row t+DEPTH as soon as it is consumed
	auto bounded = [&](int y) { return clampy(dir > 0 ? (y < y_stop ? y : y_stop) : (y > y_stop ? y : y_stop)); };
	// PRE (sharpen): the row streamed at step t is the one AHEAD of the output row, reflect(yb(t) + dir) with yb = bounded(y):
	// sharpened row yb needs gray rows yb-1, yb, yb+1, kept in a 4-slot ring GR (slot = step mod 4; 4 divides RING, so roles stay static)
	auto ahead = [&](int t) {
		const int a = bounded(y_first + dir * t) + dir;
		return a < 0 ? 1 : (a >= IMG ? IMG - 2 : a);   // BORDER_REFLECT_101 of the sharpen filter
	};
	uint32_t GR[4][8];
	if (PRE) {
		static_assert(!PRE || RING % 4 == 0, "gray ring period must divide the box ring");
		// slots 3 and 0 = the rows behind and at the first output row
		const int y0 = clampy(y_first), yb = y0 - dir;
		uint32_t d0[12], d1[12], T[16];
		load_row48(row_ptr(frame, yb < 0 ? 1 : (yb >= IMG ? IMG - 2 : yb)), lane, d0);
		load_row48(row_ptr(frame, y0), lane, d1);
#pragma unroll
		for (int k = 0; k < DEPTH; ++k) load_row48(row_ptr(frame, ahead(k)), lane, buf[k]);
		gray16_T(d0, T); pairs_from_T(T, GR[3]);
		gray16_T(d1, T); pairs_from_T(T, GR[0]);
#pragma unroll
		for (int i = 0; i < 8; ++i) { GR[1][i] = 0; GR[2][i] = 0; }
	} else {
#pragma unroll
		for (int k = 0; k < DEPTH; ++k) load_row48(row_ptr(frame, clampy(y_first + dir * k)), lane, buf[k]);
	}

	// Straight-line body: every unrolled step runs unconditionally (rows past the strip are clamped re-reads whose results
	// are never stored), so the refill of a consumed buffer is an unconditional load into the same registers -- no phi,
	// no copy, and the compiler's vmcnt waits only ever cover the oldest row in flight.
	for (int t0 = 0; t0 < total; t0 += RING) {
#pragma unroll
		for (int s = 0; s < RING; ++s) {
			const int t = t0 + s;
			const int y = y_first + dir * t;
			uint32_t (&d)[12] = buf[s % DEPTH];
			uint32_t G[8];
			int yl = y;   // the image row held in d
			if (PRE) {
				yl = ahead(t);
				uint32_t T[16];
				gray16_T(d, T);
				pairs_from_T(T, GR[(s + 1) & 3]);
				sharp_pairs(GR[(s + 3) & 3], GR[s & 3], GR[(s + 1) & 3], G);
				if (bounded(y + dir) == bounded(y)) {
					// the output row repeats (replicated border of the threshold source, or past the strip): re-seat the two
					// rows behind so the next step sees the same neighbourhood. Only the image's first / last strip gets here.
#pragma unroll
					for (int i = 0; i < 8; ++i) { GR[(s + 1) & 3][i] = GR[s & 3][i]; GR[s & 3][i] = GR[(s + 3) & 3][i]; }
				}
			} else {
				uint32_t T[16];
				gray16_T(d, T);
				pairs_from_T(T, G);
			}

			// colour column sums: rows 9r+9 .. 9r+14 are the inner rows of cell row r (cell top = 8 + 9r)
			const bool in_grid = yl >= y_begin && yl < y_end && yl >= OFFSET + 1 && yl < OFFSET + DIM * PITCH;
			const int ph = in_grid ? (yl - OFFSET) % PITCH : 0;
			if (ph >= 1 && ph <= 6) {
#pragma unroll
				for (int k = 0; k < 12; ++k) {
					acc_e[k] += d[k] & 0x00FF00FFu;
					acc_o[k] += __builtin_amdgcn_perm(0u, d[k], 0x0c030c01u);   // (d >> 8) & 0x00FF00FF in one op
				}
			}
			// d is dead: refill it with the row DEPTH steps ahead
			load_row48(row_ptr(frame, PRE ? ahead(t + DEPTH) : bounded(y + dir * DEPTH)), lane, d);

			if (ph == (dir > 0 ? 6 : 1)) {   // the cell row's last inner row in walking order
				uint16_t* sc = s_col[wave];
				uint2* dst = reinterpret_cast<uint2*>(sc + 48 * lane);
#pragma unroll
				for (int k = 0; k < 12; ++k) {
					uint2 v;
					v.x = (acc_e[k] & 0xFFFFu) | (acc_o[k] << 16);            // bytes 0,1 of dword k
					v.y = (acc_e[k] >> 16) | (acc_o[k] & 0xFFFF0000u);        // bytes 2,3
					dst[k] = v;
					acc_e[k] = 0; acc_o[k] = 0;
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				const int cell_row = (yl - OFFSET) / PITCH;
#pragma unroll
				for (int half = 0; half < 2; ++half) {
					const int c = lane + 64 * half;
					if (c < DIM) {
						const uint16_t* src = sc + 27 * c + 27;   // byte 3*(9c+9) of the row
						uint32_t r = 0, gg = 0, b = 0;
#pragma unroll
						for (int k = 0; k < 6; ++k) { r += src[3 * k]; gg += src[3 * k + 1]; b += src[3 * k + 2]; }
						cm[cell_row * DIM + c] = (r / 36u) | ((gg / 36u) << 8) | ((b / 36u) << 16);
					}
				}
				__builtin_amdgcn_wave_barrier();
			}

			uint32_t bits = 0;
			switch (s) {   // SLOT must be a compile-time constant; `s` is one after unrolling
				case 0: bits = box_row<RAD, 0>(ring, C, G, true); break;
				case 1: bits = box_row<RAD, 1>(ring, C, G, true); break;
				case 2: bits = box_row<RAD, 2>(ring, C, G, true); break;
				case 3: bits = box_row<RAD, 3>(ring, C, G, true); break;
				case 4: bits = box_row<RAD, 4>(ring, C, G, true); break;
				case 5: bits = box_row<RAD, 5>(ring, C, G, true); break;
				case 6: bits = box_row<RAD, 6 % RING>(ring, C, G, true); break;
				default: bits = box_row<RAD, 7 % RING>(ring, C, G, true); break;
			}
			{
				// gather the 8 x 16 bits of lanes 8k..8k+7 into lane 8k and store 16 bytes (columns 128k .. 128k+127 of the row)
				const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0xB1, 0xf, 0xf, true);   // lane^1
				const uint32_t w0 = (lane & 1) ? ((nb << 16) | bits) : ((bits << 16) | nb);                        // word of lanes (2j,2j+1), in both
				const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x4E, 0xf, 0xf, true);      // quad_perm:[2,3,0,1]: the quad's other word
				// lanes 4q..4q+3: lane 4q has w0 = word(2q), w1 = word(2q+1). Bring the next quad's two words over (row_shl:4).
				const uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x104, 0xf, 0xf, true);     // row_shl:4 -> lane l gets lane l+4
				const uint32_t w3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x104, 0xf, 0xf, true);
				if (t >= 2 * RAD && t < total && !(lane & 7)) {
					uint4 v; v.x = w0; v.y = w1; v.z = w2; v.w = w3;
					*reinterpret_cast<uint4*>(out + (size_t)(y - dir * RAD) * 32 + (lane >> 3) * 4) = v;   // the row whose window just completed
				}
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------ symbol matching
// 10x10 bit window whose top-left pixel is (x0, y0): ten 10-bit rows, MSB = leftmost (average_hash.h:63-75)
__device__ __forceinline__ void window_rows(const uint32_t* __restrict__ plane, int x0, int y0, uint32_t rows[10])
{
	const int j = x0 >> 5, sh = 54 - (x0 & 31);
	const int j1 = j + 1 > 31 ? 31 : j + 1;   // when j == 31 the window ends inside word 31 (x0 <= 1013)
#pragma unroll
	for (int i = 0; i < 10; ++i) {
		const uint32_t* r = plane + (size_t)(y0 + i) * 32;
		uint64_t v = ((uint64_t)r[j] << 32) | r[j1];
		rows[i] = (uint32_t)(v >> sh) & 0x3FFu;
	}
}

// ahash_result.h:70-106 / bit_extractor.h:23-51: window id w = 8x8 block at column w%3, row w/3 of the 10x10 window
__device__ __forceinline__ uint64_t window_hash(const uint32_t rows[10], int w)
{
	const int cs = 2 - (w % 3), r0 = w / 3;
	uint64_t h = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) h = (h << 8) | ((rows[r0 + k] >> cs) & 0xFFu);
	return h;
}

// best (distance << 4 | tile) over the 16 tiles for one hash; first minimum wins (CimbDecoder.cpp:111-131)
__device__ __forceinline__ uint32_t best_tile(uint64_t h)
{
	uint32_t best = 0xFFFFu;
#pragma unroll
	for (int t = 0; t < 16; ++t) {
		uint32_t d = (uint32_t)__popcll(h ^ c_tile[t]);
		uint32_t key = (d << 4) | (uint32_t)t;
		best = key < best ? key : best;
	}
	return best;
}

__device__ __forceinline__ bool is_seed(int i)
{
	// FloodDecodePositions.cpp:27-41
	return i == 0 || i == TOP_W - 1 || i == NCELLS - 1 || i == NCELLS - TOP_W || i == TOP_CELLS || i == TOP_CELLS + DIM - 1 ||
	       i == NCELLS - 1 - TOP_CELLS || i == NCELLS - TOP_CELLS - DIM;
}

// Exact-match shortcut: on clean frames ~99 % of the centre hashes ARE one of the 16 tile hashes. A 5-bit perfect hash of
// the 64-bit value ((lo ^ hi) * 0x61b91 >> 27 is collision-free over the 16 tiles) names the only tile it could be; one
// 64-bit compare confirms distance 0. Anything else takes the full popcount match below.
__constant__ uint8_t c_tile_slot[32] = {16, 9, 13, 16, 1, 0, 16, 16, 8, 2, 3, 16, 16, 7, 16, 16, 16, 16, 6, 12, 16, 16, 11, 16, 10, 5, 16, 4, 15, 16, 16, 14};
__device__ __forceinline__ uint32_t exact_tile(uint64_t h)
{
	const uint32_t slot = c_tile_slot[(((uint32_t)h ^ (uint32_t)(h >> 32)) * 0x61b91u) >> 27];
	return (slot < 16 && c_tile[slot] == h) ? slot : 16u;
}

// K2: every cell evaluated at drift (0,0). If, for every cell, no shifted window beats the centre one (4 side windows
// for ordinary cells, all 8 for the flood seeds, which may be popped in 9-window mode), the reference's flood visits
// every cell at drift (0,0) with cooldown 4|0xFE whatever its heap order, so symbol = argmin_tile popcnt(centre ^ tile)
// exactly (DESIGN.md "fast path"). Otherwise flag the frame for K2b.
// One lane per cell for the centre match (the common case ends there: distance 0). The few cells with a non-zero centre
// distance are then re-examined one at a time by the whole wave: their window rows are broadcast, the 4 (+4) shifted
// hashes are formed once on uniform data and the 64 lanes split the (window, tile) pairs.
// 10x10 window rows out of an LDS copy of the bit rows (32 words per row)
__device__ __forceinline__ void window_rows_lds(const uint32_t* rows32, int x0, uint32_t rows[10])
{
	const int j = x0 >> 5, sh = 54 - (x0 & 31);
	const int j1 = j + 1 > 31 ? 31 : j + 1;
#pragma unroll
	for (int i = 0; i < 10; ++i) {
		const uint32_t* r = rows32 + i * 32;
		uint64_t v = ((uint64_t)r[j] << 32) | r[j1];
		rows[i] = (uint32_t)(v >> sh) & 0x3FFu;
	}
}

// linear cell index of grid cell (row, col), or -1 inside the corner cut-outs (CellPositions.cpp:5-51)
__device__ __forceinline__ int cell_index(int row, int col)
{
	const bool margin = row < MARKER || row >= DIM - MARKER;
	if (margin && (col < MARKER || col >= DIM - MARKER)) return -1;
	if (row < MARKER) return row * TOP_W + (col - MARKER);
	if (row < DIM - MARKER) return TOP_CELLS + (row - MARKER) * DIM + col;
	return TOP_CELLS + MID_CELLS + (row - (DIM - MARKER)) * TOP_W + (col - MARKER);
}

// One workgroup per 16 cell rows: the 145 bit rows they touch (18.6 KB) are staged in LDS with coalesced loads. Few, fat
// waves on purpose: with one short wave per 64 cells the kernel was bound by the workgroup dispatch rate.
// Phase 1 (every cell, cheap): centre hash from 8 bit rows, exact-match shortcut; a cell that is not an exact tile is queued.
// Phase 2 (queued cells only, ~1 % on clean frames, all of them on noisy ones): the full popcount match of the centre and,
// if its distance is not 0, of the 4 (seeds: 8) shifted windows -- one lane per queued cell, so the rare work is compacted
// into a few lanes instead of stalling every wave.
constexpr int K2_BLOCK_ROWS = 16, K2_CELLS = K2_BLOCK_ROWS * DIM;   // 1792 cells per workgroup
__global__ __launch_bounds__(256) void k_symbols(const uint32_t* __restrict__ plane, Tables tb, uint8_t* __restrict__ symbols,
                                                 uint32_t* __restrict__ flood_flag, int f0)
{
	__shared__ uint32_t s_rows[(K2_BLOCK_ROWS * PITCH + 1) * 32];
	__shared__ uint16_t s_queue[K2_CELLS];
	__shared__ int s_qn;
	// the exact-match tables next to the data: per-lane lookups in __constant__ memory are two dependent vector loads per cell
	__shared__ uint64_t s_tile[16];
	__shared__ uint8_t s_slot[32];
	if (threadIdx.x < 16) s_tile[threadIdx.x] = c_tile[threadIdx.x];
	if (threadIdx.x < 32) s_slot[threadIdx.x] = c_tile_slot[threadIdx.x];
	const int f = f0 + blockIdx.y;
	const int brow0 = blockIdx.x * K2_BLOCK_ROWS;
	const uint32_t* pl = plane + (size_t)f * PLANE_WORDS + (size_t)(OFFSET + brow0 * PITCH - 1) * 32;   // first bit row needed: y0 - 1
	if (threadIdx.x == 0) s_qn = 0;
	for (int k = threadIdx.x; k < (K2_BLOCK_ROWS * PITCH + 1) * 32; k += 256) s_rows[k] = pl[k];
	__syncthreads();

	for (int lc = threadIdx.x; lc < K2_CELLS; lc += 256) {   // lc = local cell: row-in-block * 112 + col
		const int rsel = lc / DIM, col = lc % DIM;
		const int i = cell_index(brow0 + rsel, col);
		if (i < 0) continue;
		// centre 8x8 = window rows 1..8, bits 1..8 of each 10-bit row
		const uint32_t* r = s_rows + (rsel * PITCH + 1) * 32;
		const int x0 = OFFSET + col * PITCH;                   // centre block starts one pixel right of the window origin
		const int j = x0 >> 5, sh = 56 - (x0 & 31);
		const int j1 = j + 1 > 31 ? 31 : j + 1;
		uint32_t hi = 0, lo = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			uint64_t v = ((uint64_t)r[k * 32 + j] << 32) | r[k * 32 + j1];
			hi = (hi << 8) | ((uint32_t)(v >> sh) & 0xFFu);
			uint64_t w = ((uint64_t)r[(k + 4) * 32 + j] << 32) | r[(k + 4) * 32 + j1];
			lo = (lo << 8) | ((uint32_t)(w >> sh) & 0xFFu);
		}
		const uint32_t slot = s_slot[((lo ^ hi) * 0x61b91u) >> 27];
		const uint64_t cand_tile = s_tile[slot & 15u];
		if (slot < 16 && cand_tile == (((uint64_t)hi << 32) | lo)) {
			symbols[(size_t)f * NCELLS + i] = (uint8_t)slot;
		} else {
			s_queue[atomicAdd(&s_qn, 1)] = (uint16_t)lc;
		}
	}
	__syncthreads();

	bool shifted = false;
	const int qn = s_qn;
	for (int q = threadIdx.x; q < qn; q += 256) {
		const int lc = s_queue[q];
		const int rsel = lc / DIM, col = lc % DIM;
		const int i = cell_index(brow0 + rsel, col);
		uint32_t rows[10];
		window_rows_lds(s_rows + rsel * PITCH * 32, OFFSET + col * PITCH - 1, rows);
		const uint32_t centre = best_tile(window_hash(rows, 4));
		const uint32_t dc = centre >> 4;
		symbols[(size_t)f * NCELLS + i] = (uint8_t)(centre & 15u);
		if (dc != 0) {
			uint32_t other = 0xFFFFu;
			const int side[4] = {5, 7, 3, 1};
#pragma unroll
			for (int k = 0; k < 4; ++k) { uint32_t bt = best_tile(window_hash(rows, side[k])); other = bt < other ? bt : other; }
			if (is_seed(i)) {
				const int corner[4] = {8, 0, 2, 6};
#pragma unroll
				for (int k = 0; k < 4; ++k) { uint32_t bt = best_tile(window_hash(rows, corner[k])); other = bt < other ? bt : other; }
			}
			shifted |= (other >> 4) < dc;
		}
	}
	if (shifted) atomicOr(&flood_flag[f], 1u);
}

// ------------------------------------------------------------------------------------------------ K2b exact flood
// FloodDecodePositions.cpp:17-134 + CimbReader.cpp:139-162 + CimbDecoder.cpp:101-147, literally: the visiting order of the
// reference's std::priority_queue decides which neighbour's drift a cell inherits, so the queue is replayed operation for
// operation (libstdc++ bits/stl_heap.h push_heap / pop_heap, including how equal priorities fall) -- a serial algorithm.
// One wavefront per flagged frame, everything it touches per step in LDS (the frame's bit plane, per-cell state, the heap),
// and the 64 lanes cooperate inside every step:
//   * heap pop: the walk down the tree fetches a whole 6-level subtree with one gather (lane l <- node r*2^d + l, d = depth
//     of l in the subtree), picks children with v_readlane, and the libstdc++ "hole" shuffle becomes one scattered write;
//   * heap push: the <=17 ancestors of the new slot are fetched with one gather, a ballot finds where the climb stops;
//   * decode: the (window, tile) popcounts are spread over the lanes, DPP min-reduce;
//   * the <=12 neighbour offers (4 adjacent + 8 "horizon") are evaluated one per lane from a precomputed candidate table.
// A heap entry carries everything the reference keeps in _instructions[] for a cell (an accepted offer always has a
// strictly smaller priority than the cell's previous ones, so the entry that pops first IS the latest instruction):
//   prio(7) << 25 | cell(14) << 11 | dx+8 (4) << 7 | dy+8 (4) << 3 | cooldown code (3)
// The eight seed cells are the exception (their seed entry pops with priority 0/1 but must use the latest accepted offer, if
// any): their current instruction lives in s_seed[]. s_state[cell] (u8) = best offered priority + 1, 0x7F (no offer yet), or, once
// visited, 0x80 | symbol. The symbols go out when the frame is done; the position the colour pass reads is one fire-and-forget
// 2-byte store per step (nothing in the loop waits for stores: the only loads are the two prefetches, consumed a pop later).
#ifndef CIMBAR_HEAP_LDS
#define CIMBAR_HEAP_LDS 10240
#endif
constexpr int HEAP_LDS = CIMBAR_HEAP_LDS;  // heap slots held in LDS (40 KiB; a clean shifted frame peaks near 9 200 live entries); deeper slots
                                 // spill to the global scratch under the same indices (tests build a second library with
                                 // CIMBAR_HEAP_LDS=1024 so that the spill path runs on ordinary frames)
struct FloodScratch {
	uint32_t* heap;      // [FLOOD_GRID][HEAP_CAP] one spill area per workgroup; only indices >= HEAP_LDS are ever touched
};

__device__ __forceinline__ uint32_t cool_enc(uint32_t c) { return c == 0xFEu ? 0u : (c == 0xFFu ? 2u : c); }   // real values: 1,3,4,5,7
__device__ __forceinline__ uint32_t cool_dec(uint32_t k) { return k == 0u ? 0xFEu : (k == 2u ? 0xFFu : k); }

typedef __attribute__((address_space(3))) uint32_t lds_u32;
// The spill half of the heap (slots >= HEAP_LDS, global memory), only compiled into the SPILL = true instance of a step
// (taken when the heap is within a step's growth of HEAP_LDS). Out of line on purpose: it is cold, and inlined the compiler
// would have to assume pending global loads around every heap access.
__device__ __attribute__((noinline)) uint32_t heap_spill_get(const uint32_t* g, int i, uint32_t v) { return i >= HEAP_LDS ? g[i] : v; }
__device__ __attribute__((noinline)) void heap_spill_set(uint32_t* g, int i, uint32_t v, bool on) { if (on && i >= HEAP_LDS) g[i] = v; }

struct PopPrefetch { uint32_t value, v, vl, vr; };

struct WaveHeap {
	lds_u32* lds;
	uint32_t* glob;
	int n;
	// i, on: per lane. Reads are unconditional (a masked-off lane reads slot 0) so that they cost no branch.
	template <bool SPILL>
	__device__ __forceinline__ uint32_t get(int i, bool on = true) const
	{
		uint32_t v = lds[(on && i < HEAP_LDS) ? i : 0];
		if constexpr (SPILL) v = heap_spill_get(glob, on ? i : 0, v);
		return v;
	}
	template <bool SPILL>
	__device__ __forceinline__ void set(int i, uint32_t v, bool on) const
	{
		if (on && i < HEAP_LDS) lds[i] = v;
		if constexpr (SPILL) heap_spill_set(glob, i, v, on);
	}

	// std::push_heap after push_back (stl_heap.h __push_heap): climb while the parent's priority is greater
	template <bool SPILL>
	__device__ __forceinline__ void push(uint32_t e, int lane)
	{
		const int pos = n++;
		const int depth = 31 - __builtin_clz((unsigned)pos + 1u);        // ancestors a_1 .. a_depth (= root)
		const int a = (int)(((unsigned)pos + 1u) >> (lane < 31 ? lane : 31)) - 1;   // a_0 = pos
		const bool anc = lane >= 1 && lane <= depth;
		const uint32_t v = get<SPILL>(a, anc);
		const unsigned long long stop = __ballot(anc && !((v >> 25) > (e >> 25)));
		const int m = stop ? (int)__builtin_ctzll(stop) - 1 : depth;   // the new element lands in a_m
		const uint32_t vnext = from_right_lane(v, 0u);                   // lane k <- H[a_{k+1}]
		set<SPILL>(a, lane < m ? vnext : e, lane <= m);
	}

	// what pop() reads first, so that it can be requested together with the peek at the top
	template <bool SPILL>
	__device__ __forceinline__ PopPrefetch prefetch(int lane) const
	{
		const int len = n - 1, half = (len - 1) / 2;
		PopPrefetch p;
		p.value = get<SPILL>(len > 0 ? len : 0);
		p.v = get<SPILL>(lane, lane < 63 && lane < len);
		const bool inner = lane < 31 && lane < half;
		p.vl = get<SPILL>(2 * lane + 1, inner);
		p.vr = get<SPILL>(2 * lane + 2, inner);
		return p;
	}

	// std::pop_heap + pop_back (stl_heap.h __pop_heap -> __adjust_heap -> __push_heap); the caller has read the top already.
	// __adjust_heap walks the hole from the root to the bottom, always into the child the comparator prefers (the right one on
	// ties), moving that child up; __push_heap then lifts the old last element `value` back up over the moved elements. In
	// terms of the OLD array and the path c_0 = root, c_1, .., c_L: with j = the deepest k >= 1 whose H[c_k] has priority <=
	// value's (0 if none), H[c_{k-1}] <- H[c_k] for k <= j, H[c_j] <- value, everything else stays.
	// The walk takes five levels per LDS round trip: lane l fetches node l of the 63-node subtree under the hole (global
	// index hole*2^d + l, d = depth of l) and, if it is an inner node, its two children; a ballot turns the comparisons into
	// one bit per node and the descent is five scalar bit-lookups.
	template <bool SPILL>
	__device__ __forceinline__ void pop(int lane, const PopPrefetch& pf)
	{
		const int len = n - 1;                  // elements that remain; `value` = the old last one, re-inserted from the root
		if (len == 0) { n = 0; return; }
		const uint32_t value = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.value);
		const uint32_t vprio = value >> 25;
		const int half = (len - 1) / 2;         // nodes below this index have two children
		const int d = 31 - __builtin_clz((unsigned)lane + 1u);
		uint32_t bv[4];
		int bg[4];
		bool bp[4];
		int S = 1, depth = 0, j = 0;            // 1-based index of the hole, its depth, and the j of the comment above
#pragma unroll
		for (int B = 0; B < 4; ++B) {
			bv[B] = 0; bg[B] = 0; bp[B] = false;
			if (S - 1 < half) {
				const int gi = ((S - 1) << d) + lane;
				const bool inner = lane < 31 && gi < half;
				uint32_t v, vl, vr;
				if (B == 0) { v = pf.v; vl = pf.vl; vr = pf.vr; }
				else {
					v = get<SPILL>(gi, lane < 63 && gi < len);
					vl = get<SPILL>(2 * gi + 1, inner);
					vr = get<SPILL>(2 * gi + 2, inner);
				}
				// comp(first[second], first[second-1]) -> second-- : the left child only if the right one's priority is greater
				const unsigned long long right = __ballot(inner && !((vr >> 25) > (vl >> 25)));
				unsigned long long pm = 0;
				int m = 1;                            // 1-based local node
#pragma unroll
				for (int lev = 0; lev < 5; ++lev) {
					if (S - 1 < half) {
						const int t = (int)((right >> (m - 1)) & 1ull);
						m = 2 * m + t;
						S = 2 * S + t;
						++depth;
						pm |= 1ull << (m - 1);
					}
				}
				const bool onpath = (pm >> lane) & 1ull;
				bv[B] = v; bg[B] = gi; bp[B] = onpath;
				const unsigned long long cm = __ballot(onpath && (v >> 25) <= vprio);
				if (cm) j = 5 * B + (31 - __builtin_clz((unsigned)(63 - (int)__builtin_clzll(cm)) + 1u));
			}
		}
		// (four blocks cover 20 levels; HEAP_CAP < 2^18)
		int tail = -1;                              // a last node with a left child only
		uint32_t tailv = 0;
		if ((len & 1) == 0 && S - 1 == (len - 2) / 2) {
			tail = 2 * (S - 1) + 1;
			tailv = (uint32_t)__builtin_amdgcn_readfirstlane((int)get<SPILL>(tail));
			++depth;
			if ((tailv >> 25) <= vprio) j = depth;
		}
#pragma unroll
		for (int B = 0; B < 4; ++B) {
			if (B == 0 || 5 * B < depth) {
				const int dp = 5 * B + d;
				set<SPILL>((bg[B] - 1) >> 1, bv[B], bp[B] && dp <= j);
				set<SPILL>(bg[B], value, bp[B] && dp == j);
			}
		}
		if (tail >= 0 && j == depth) {
			set<SPILL>((tail - 1) >> 1, tailv, lane == 0);
			set<SPILL>(tail, value, lane == 0);
		}
		if (j == 0) set<SPILL>(0, value, lane == 0);
		n = len;
	}
};

__device__ __forceinline__ uint32_t calc_cooldown(uint32_t previous, uint32_t idx)
{
	// CellDrift.cpp:33-43
	if (idx == 4) return 4;
	if ((idx & 1) == 0) return 0xFF;
	if (((previous ^ idx) & 0xFF) == 6) return 0xFF;
	return idx;
}

// minimum of a 32-bit value over the wave (uniform result)
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
	uint32_t o;
	o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = o < v ? o : v;    // quad_perm:[1,0,3,2]
	o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = o < v ? o : v;    // quad_perm:[2,3,0,1]
	o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = o < v ? o : v;   // row_half_mirror
	o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false); v = o < v ? o : v;   // row_mirror
	const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
	const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), e = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
	const uint32_t ab = a < b ? a : b, ce = c < e ? c : e;
	return ab < ce ? ab : ce;
}

// FloodDecodePositions.cpp:27-41, in push order; -1 for everything else. All eight sit in the first 712 or last 712 cells.
__device__ __forceinline__ int seed_slot(int i)
{
	if ((unsigned)(i - (TOP_CELLS + DIM)) < (unsigned)(NCELLS - 2 * (TOP_CELLS + DIM))) return -1;
	int s = -1;
	s = i == 0 ? 0 : s;
	s = i == TOP_W - 1 ? 1 : s;
	s = i == NCELLS - 1 ? 2 : s;
	s = i == NCELLS - TOP_W ? 3 : s;
	s = i == TOP_CELLS ? 4 : s;
	s = i == TOP_CELLS + DIM - 1 ? 5 : s;
	s = i == NCELLS - 1 - TOP_CELLS ? 6 : s;
	s = i == NCELLS - TOP_CELLS - DIM ? 7 : s;
	return s;
}

constexpr int FLOOD_GRID = 768;   // state + heap = 52 KiB of LDS: three workgroups per CU
__global__ __launch_bounds__(64) void k_flood(const uint32_t* __restrict__ plane, Tables tb, FloodScratch sc,
                                              const uint32_t* __restrict__ flood_flag, uint8_t* __restrict__ symbols,
                                              int8_t* __restrict__ drift, int f0, int nframes, int area0)
{
	__shared__ __attribute__((aligned(16))) uint8_t s_state[NCELLS + 16];
	__shared__ uint32_t s_heap[HEAP_LDS];
	__shared__ uint32_t s_seed[8];                                           // prio << 16 | (dx+8) << 7 | (dy+8) << 3 | cooldown code
	const int lane = threadIdx.x;
	constexpr uint32_t DEFAULT_D = (8u << 7) | (8u << 3) | 0u;              // CellDrift(), cooldown 0xFE

	for (int fi = blockIdx.x; fi < nframes; fi += gridDim.x) {
		const int f = f0 + fi;
		if (!flood_flag[f]) continue;
		const uint32_t* pl = plane + (size_t)f * PLANE_WORDS;
		__syncthreads();   // the previous frame of this workgroup is completely done with the LDS
		{
			uint4 ff; ff.x = ff.y = ff.z = ff.w = 0x7F7F7F7Fu;
			for (int i = lane; i < (NCELLS + 16) / 16; i += 64) reinterpret_cast<uint4*>(s_state)[i] = ff;
			if (lane < 8) s_seed[lane] = (0xFEu << 16) | DEFAULT_D;
		}
		WaveHeap hp{(lds_u32*)s_heap, sc.heap + (size_t)(area0 + blockIdx.x) * HEAP_CAP, 0};   // spill scratch belongs to the workgroup, not the frame
		__syncthreads();
		{
			const uint32_t last = NCELLS - 1;
			auto seed = [&](uint32_t prio, uint32_t cell) { hp.push<false>((prio << 25) | (cell << 11) | DEFAULT_D, lane); };
			seed(0, 0u); seed(0, (uint32_t)(TOP_W - 1)); seed(0, last); seed(0, last - (TOP_W - 1));
			seed(1, (uint32_t)TOP_CELLS); seed(1, (uint32_t)(TOP_CELLS + DIM - 1)); seed(1, last - TOP_CELLS); seed(1, last - (TOP_CELLS + DIM - 1));
		}

		constexpr uint32_t ORDER8 = 0x20813754u;   // nibble k = k-th window visited: 4,5,7,3,1,8,0,2 and then 6 (ahash_result.h:26)
		const uint64_t my_tile = c_tile[lane & 15];
		// lanes 0..8 each build the 8x8 hash of window w = lane out of the ten 10-bit rows (bit_extractor.h:23-51)
		const int hw = lane < 9 ? lane : 0;
		const uint32_t h_cs = 2u - (uint32_t)(hw % 3), h_r8 = 8u * (uint32_t)(hw / 3);
#ifdef FLOOD_PROF
		unsigned long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pc0;
#define PROF_T0() pc0 = __builtin_readcyclecounter()
#define PROF_ADD(k) do { unsigned long long pc1 = __builtin_readcyclecounter(); pt[k] += pc1 - pc0; pc0 = pc1; } while (0)
#else
#define PROF_T0()
#define PROF_ADD(k)
#endif
		// one step = one decoded cell; instantiated without and with the heap's spill half
		auto step = [&](auto spill_tag) __attribute__((always_inline)) -> bool {
			constexpr bool SPILL = decltype(spill_tag)::value;
			PROF_T0();
			// ---- FloodDecodePositions::next(): pop until a cell that still needs decoding turns up. The top of the heap names
			// the cell, and (with s_seed) where its window is, before the pop has run: the twenty bit-plane words of the
			// window and the cell's offer list are requested from L2 first and arrive behind the pop.
			int i = -1;
			uint32_t e = 0, di = 0, prev_prio = 0, pw = 0, sh = 0;
			int16_t cand = -1;
			while (hp.n > 0) {
				const uint32_t topv = hp.get<SPILL>(0);
				const PopPrefetch pf = hp.prefetch<SPILL>(lane);   // same LDS round trip as the peek
				e = (uint32_t)__builtin_amdgcn_readfirstlane((int)topv);
				PROF_ADD(7);
				const int idx = (int)((e >> 11) & 0x3FFFu);
				const bool fresh = ((uint32_t)__builtin_amdgcn_readfirstlane((int)s_state[idx]) & 0x80u) == 0;
				if (fresh) {
					di = e & 0x7FFu; prev_prio = e >> 25;
					const int slot = seed_slot(idx);
					if (slot >= 0) { const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_seed[slot]); di = sv & 0x7FFu; prev_prio = sv >> 16; }
					// CellPositions.cpp:5-51 (the three bands of mode B)
					int cx, cy;
					if (idx < TOP_CELLS) { cx = OFFSET + (MARKER + idx % TOP_W) * PITCH; cy = OFFSET + (idx / TOP_W) * PITCH; }
					else if (idx < TOP_CELLS + MID_CELLS) { const int q = idx - TOP_CELLS; cx = OFFSET + (q % DIM) * PITCH; cy = OFFSET + (MARKER + q / DIM) * PITCH; }
					else { const int q = idx - TOP_CELLS - MID_CELLS; cx = OFFSET + (MARKER + q % TOP_W) * PITCH; cy = OFFSET + (DIM - MARKER + q / TOP_W) * PITCH; }
					const int x0 = cx + (int)(di >> 7) - 9, y0 = cy + (int)((di >> 3) & 15u) - 9;   // top-left of the 10x10 window
					const int j = x0 >> 5, j1 = j + 1 > 31 ? 31 : j + 1;   // j == 31: the window ends inside word 31
					sh = 54u - (uint32_t)(x0 & 31);
					// lane 2r: first word of window row r, lane 2r+1: the next word
					if (lane < 20) pw = pl[(y0 + (lane >> 1)) * 32 + ((lane & 1) ? j1 : j)];
					if (lane < 12) cand = tb.cand[(size_t)idx * 12 + lane];
				}
				PROF_ADD(8);
				hp.pop<SPILL>(lane, pf);
#ifdef FLOOD_PROF
				pt[4] += 1;
				if ((unsigned long long)hp.n > pt[6]) pt[6] = hp.n;
#endif
				if (!fresh) continue;
				i = idx;
				break;
			}
			if (i < 0) return false;
			PROF_ADD(0);
			const int ddx = (int)(di >> 7) - 8, ddy = (int)((di >> 3) & 15u) - 8;
			const uint32_t cooldown = cool_dec(di & 7u);

			// even lanes < 20 form their 10-bit row (MSB = leftmost pixel); the rows become wave-uniform through readlane, and
			// lane w < 9 assembles window w's hash: byte k = bits [cs, cs+8) of row r0+k, r0 = w/3, cs = 2 - w%3
			//   -> the 10 bytes packed big-endian, cut at byte r0
			uint32_t hlo, hhi;
			{
				const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pw, 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]: lane^1
				const uint32_t row = (uint32_t)((((uint64_t)pw << 32) | nxt) >> sh);
				uint32_t tb_[10];
#pragma unroll
				for (int r = 0; r < 10; ++r) tb_[r] = ((uint32_t)__builtin_amdgcn_readlane((int)row, 2 * r) >> h_cs) & 0xFFu;
				const uint32_t W0 = (tb_[0] << 24) | (tb_[1] << 16) | (tb_[2] << 8) | tb_[3];
				const uint32_t W1 = (tb_[4] << 24) | (tb_[5] << 16) | (tb_[6] << 8) | tb_[7];
				const uint32_t W2 = (tb_[8] << 24) | (tb_[9] << 16);
				const uint64_t A = (((uint64_t)W0 << 32) | W1) << h_r8;
				hhi = (uint32_t)(A >> 32);
				hlo = (uint32_t)A | ((W2 >> 16) >> (16u - h_r8));
			}

			// (window position k in visiting order, tile t) pairs over the lanes; key = dist << 8 | k << 4 | t, min wins
			const int nwin = (cooldown == 0xFEu) ? 9 : 5;         // CimbDecoder.cpp:144
			uint32_t best = 0xFFFFFFFFu;
#pragma unroll
			for (int it = 0; it < 3; ++it) {
				if (it * 4 >= nwin) break;
				const int k = (lane >> 4) + 4 * it;
				const uint32_t w = k < 8 ? (ORDER8 >> (4 * k)) & 15u : 6u;
				const uint32_t wlo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)hlo);
				const uint32_t whi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)hhi);
				const bool skip = k >= nwin || (w == cooldown && w != 4u);   // CimbDecoder.cpp:114-115
				const uint32_t d = (uint32_t)__popc(wlo ^ (uint32_t)my_tile) + (uint32_t)__popc(whi ^ (uint32_t)(my_tile >> 32));
				const uint32_t key = skip ? 0xFFFFFFFFu : ((d << 8) | ((uint32_t)k << 4) | (uint32_t)(lane & 15));
				best = key < best ? key : best;
			}
			best = wave_min(best);
			PROF_ADD(1);

			const uint32_t kbest = (best >> 4) & 15u;
			const uint32_t error_distance = best >> 8, w = kbest < 8 ? (ORDER8 >> (4 * kbest)) & 15u : 6u, bits = best & 15u;
			const int bdx = (int)(w % 3) - 1, bdy = (int)(w / 3) - 1;                 // CellDrift.h:13-15
			int ndx = ddx + bdx, ndy = ddy + bdy;                                     // CellDrift.cpp:23-31
			ndx = ndx > 7 ? 7 : (ndx < -7 ? -7 : ndx);
			ndy = ndy > 7 ? 7 : (ndy < -7 ? -7 : ndy);
			const uint32_t ncool = calc_cooldown(cooldown, w);
			// visited + the symbol; the position the colour pass reads (CimbReader.cpp:158-160 pos.x/y) goes straight to memory
			if (lane == 0) {
				s_state[i] = (uint8_t)(0x80u | bits);
				*reinterpret_cast<uint16_t*>(drift + ((size_t)f * NCELLS + i) * 2) =
				    (uint16_t)(((uint32_t)(ddx + bdx) & 0xFFu) | (((uint32_t)(ddy + bdy) & 0xFFu) << 8));
			}

			// ---- FloodDecodePositions::update(): lanes 0-3 the adjacent cells (right, left, bottom, top), 4-7 the horizontal
			// "horizon" (:93-111), 8-11 the vertical one (:113-129), in the reference's offer order. No cell occurs twice in
			// one list (checked when the table is built), so the twelve update_adjacents checks are independent.
			const unsigned long long have = __ballot(cand >= 0);
			const bool far = prev_prio < 3 && error_distance < 3 && cooldown == 4 && ncool == 4;
			const uint32_t lanes_ok = 0xFu | (far && (have & 3ull) == 3ull ? 0xF0u : 0u) | (far && (have & 12ull) == 12ull ? 0xF00u : 0u);
			const bool want = lane < 12 && ((lanes_ok >> (lane & 15)) & 1u) && cand >= 0;
			// update_adjacents (:69-83): still to decode and strictly better than what the cell was offered before
			const uint32_t cst = s_state[want ? (int)cand : 0];
			const bool accept = want && !(cst & 0x80u) && cst >= error_distance + 2u;
			const uint32_t dcode = ((uint32_t)(ndx + 8) << 7) | ((uint32_t)(ndy + 8) << 3) | cool_enc(ncool);
			if (accept) {
				s_state[cand] = (uint8_t)(error_distance + 1u);
				const int sl = seed_slot((int)cand);
				if (sl >= 0) s_seed[sl] = (error_distance << 16) | dcode;
			}
			unsigned long long acc = __ballot(accept);
			PROF_ADD(2);
			while (acc) {
				const int q = (int)__builtin_ctzll(acc);
				acc &= acc - 1;
				const uint32_t cell = (uint32_t)__builtin_amdgcn_readlane((int)cand, q);
				hp.push<SPILL>((error_distance << 25) | (cell << 11) | dcode, lane);
#ifdef FLOOD_PROF
				pt[5] += 1;
#endif
			}
			PROF_ADD(3);
			return true;
		};
		for (int count = 0; count < NCELLS; ++count) {
			// a step pushes at most 12 entries: below that margin no heap index can reach the spill half
			const bool more = hp.n + 16 <= HEAP_LDS ? step(std::false_type{}) : step(std::true_type{});
			if (!more) break;
		}
#ifdef FLOOD_PROF
		if (lane == 0) for (int k = 0; k < 10; ++k) { sc.heap[(size_t)(area0 + blockIdx.x) * HEAP_CAP + 2 * k] = (uint32_t)pt[k]; sc.heap[(size_t)(area0 + blockIdx.x) * HEAP_CAP + 2 * k + 1] = (uint32_t)(pt[k] >> 32); }
#endif
		// symbols of the cells that were visited (all of them, unless the grid were disconnected)
		__syncthreads();
		for (int c = lane; c < NCELLS; c += 64) {
			const uint32_t v = s_state[c];
			if (v & 0x80u) symbols[(size_t)f * NCELLS + c] = (uint8_t)(v & 15u);
		}
	}
}

// ------------------------------------------------------------------------------------------------ K3 Reed-Solomon
// reed_solomon_stream.h:54-77 -> libcorrect decode.c:299-379, one 155-byte block per wavefront. The block is gathered
// straight out of the per-cell symbol (4 bit, 2 cells/byte) or colour (2 bit, 4 cells/byte) arrays through the inverse
// interleave map (Decoder.h:91-96,112; bitbuffer.h:62-84), so the 6200/3100-byte pre-RS streams never exist in memory.
// XOR of a 32-bit value over all 64 lanes (result uniform): 4 DPP butterflies inside each 16-lane row, then the four rows
__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
	v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm:[1,0,3,2]
	v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm:[2,3,0,1]
	v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
	v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);   // row_mirror
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^
	       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

struct RsShared {
	uint8_t exp[768];         // exp[512..767] = 0: where a zero byte's "logarithm" points, so that it contributes nothing
	uint8_t log[256];
	uint8_t enc[4][160];      // block in transmit order
	__attribute__((aligned(4))) uint8_t synd[4][32];
	uint8_t loc[4][72];
	uint8_t last[4][72];
	uint8_t evalr[4][32];
};

__device__ __forceinline__ uint8_t gf_mul(const RsShared& s, uint8_t l, uint8_t r) { return (!l || !r) ? 0 : s.exp[(unsigned)s.log[l] + s.log[r]]; }
__device__ __forceinline__ uint8_t gf_div(const RsShared& s, uint8_t l, uint8_t r) { return (!l || !r) ? 0 : s.exp[255u + s.log[l] - s.log[r]]; }
// value of sum_i coef[i] * e^i for i <= order (polynomial.c:113-131 with element_exp rows = successive powers of e), e != 0
__device__ __forceinline__ uint8_t gf_eval(const RsShared& s, const uint8_t* coef, int order, uint8_t e)
{
	unsigned le = s.log[e] % 255u, acc = 0;   // log[1] == 255 -> 0
	uint8_t res = 0;
	for (int i = 0; i <= order; ++i) {
		if (coef[i]) res ^= s.exp[(unsigned)s.log[coef[i]] + acc];
		acc += le; if (acc >= 255u) acc -= 255u;
	}
	return res;
}

template <int BITS>   // 4: symbol stream, 2: colour stream
__global__ __launch_bounds__(256) void k_rs(const uint8_t* __restrict__ cells, Tables tb, int f0, int nframes, int first_chunk,
                                            uint8_t* __restrict__ chunks, uint8_t* __restrict__ rs_ok, int ok_offset)
{
	constexpr int NBLK = (BITS == 4) ? SYM_BLOCKS : COL_BLOCKS;
	constexpr int PER_BYTE = 8 / BITS;
	__shared__ RsShared s;
	for (int k = threadIdx.x; k < 768; k += 256) s.exp[k] = k < 512 ? c_gf_exp[k] : (uint8_t)0;
	s.log[threadIdx.x] = c_gf_log[threadIdx.x];
	__syncthreads();

	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int gb = blockIdx.x * 4 + wv;                 // global block number over the batch
	const int b = gb % NBLK;
	if (gb / NBLK >= nframes) return;
	const int f = f0 + gb / NBLK;
	const uint8_t* cf = cells + (size_t)f * NCELLS;
	uint8_t* enc = s.enc[wv];

	// gather: stream byte B = 155*b + k packs stream cells PER_BYTE*B .. +PER_BYTE-1, first cell in the high bits.
	// Lane l owns bytes l, l+64, l+128 of the block (registers) and also parks them in LDS for the correction / output steps.
	uint32_t mine[3];
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		const int k = lane + 64 * r;
		uint32_t v = 0;
		if (k < RS_BLOCK) {
			const int sidx = (RS_BLOCK * b + k) * PER_BYTE;
#pragma unroll
			for (int q = 0; q < PER_BYTE; ++q) v = (v << BITS) | (cf[tb.stream_cell[sidx + q]] & ((1u << BITS) - 1u));
			enc[k] = (uint8_t)v;
		}
		mine[r] = v;
	}

	// syndromes S_j = r(alpha^(j+1)), r(x) = sum_i enc[154-i] x^i (decode.c:12-28), evaluated term-parallel: lane's byte k
	// contributes enc[k] * alpha^((j+1)*(154-k)); the 64 partial sums are XOR-reduced across the wave, four syndromes
	// (one per byte of a dword) at a time. No dependent chain of table look-ups, every lane busy.
	// Two roots at a time: P = (e_j, e_j+1) as packed u16, e_j = ((j+1) * (154-k)) mod 255; the next pair is P + 2*(154-k) mod 255,
	// reduced with one packed subtract + packed min (x >= 255 ? x - 255 : x  ==  min(x, x - 255) in unsigned 16-bit).
	typedef unsigned short us2 __attribute__((ext_vector_type(2)));
	uint32_t lgz[3];
	us2 P[3], step2[3];
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		const int k = lane + 64 * r;
		const uint32_t sk = k < RS_BLOCK ? (uint32_t)(RS_BLOCK - 1 - k) : 0u;       // power of x this byte multiplies, < 255
		const uint32_t s2 = 2u * sk >= 255u ? 2u * sk - 255u : 2u * sk;
		lgz[r] = mine[r] ? (uint32_t)s.log[mine[r] & 0xFFu] : 512u;
		P[r] = us2{(unsigned short)sk, (unsigned short)s2};
		step2[r] = us2{(unsigned short)s2, (unsigned short)s2};
	}
	const us2 k255 = us2{255, 255};
	uint32_t any_nonzero = 0;
#pragma unroll
	for (int jg = 0; jg < (RS_PARITY + 3) / 4; ++jg) {
		uint32_t packed = 0;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (4 * jg + 2 * h < RS_PARITY) {   // RS_PARITY is even: roots come in whole pairs
				uint32_t part0 = 0, part1 = 0;
#pragma unroll
				for (int r = 0; r < 3; ++r) {
					part0 ^= s.exp[lgz[r] + P[r].x];
					part1 ^= s.exp[lgz[r] + P[r].y];
					P[r] += step2[r];
					P[r] = __builtin_elementwise_min(P[r], P[r] - k255);
				}
				packed |= (part0 | (part1 << 8)) << (16 * h);
			}
		}
		const uint32_t red = wave_xor(packed);
		any_nonzero |= red;
		if (lane == 0) reinterpret_cast<uint32_t*>(s.synd[wv])[jg] = red;
	}
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_s_waitcnt(0);   // LDS writes above visible to the wave's later reads
	const bool all_zero = any_nonzero == 0;

	int ok = 1;
	if (!all_zero) {
		uint8_t* loc = s.loc[wv];
		uint8_t* last = s.last[wv];
		const uint8_t* synd = s.synd[wv];
		// Berlekamp-Massey, decode.c:32-118. The control flow (discrepancy test, "room for more taps" test, orders, delay) is
		// libcorrect's, statement for statement; its array loops are element-wise, so lane k carries element k (and k+64) of
		// the locator / previous locator and all elements move at once. Entries above the loops' bounds keep their stale
		// values exactly as they do in libcorrect's buffers.
		for (int k = lane; k < 72; k += 64) { loc[k] = (k == 0); last[k] = (k == 0); }
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		unsigned loc_order = 0, last_order = 0, numerrors = 0, delay = 1;
		uint8_t last_disc = 1;
		for (unsigned i = 0; i < (unsigned)RS_PARITY; ++i) {
			uint32_t part = 0;
			if (lane >= 1 && (unsigned)lane <= numerrors) part = gf_mul(s, loc[lane], synd[i - lane]);
			const uint8_t disc = (uint8_t)((synd[i] ^ wave_xor(part)) & 0xFFu);
			if (!disc) { delay++; continue; }
			if (2 * numerrors <= i) {
				// last <- (last * disc / last_disc) shifted up by `delay`; then loc <- loc - last, last <- old loc, over [0, last_order + delay]
				uint8_t nloc[2] = {0, 0}, nlast[2] = {0, 0};
#pragma unroll
				for (int hh = 0; hh < 2; ++hh) {
					const unsigned k = (unsigned)lane + 64u * hh;
					if (k < 72u) {
						const uint8_t oloc = loc[k], olast = last[k];
						const uint8_t shifted = k < delay ? (uint8_t)0
						                        : (k - delay <= last_order ? gf_div(s, gf_mul(s, last[k - delay], disc), last_disc) : olast);
						if (k <= last_order + delay) { nloc[hh] = oloc ^ shifted; nlast[hh] = oloc; }
						else { nloc[hh] = oloc; nlast[hh] = shifted; }
					}
				}
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_s_waitcnt(0);
#pragma unroll
				for (int hh = 0; hh < 2; ++hh) {
					const unsigned k = (unsigned)lane + 64u * hh;
					if (k < 72u) { loc[k] = nloc[hh]; last[k] = nlast[hh]; }
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				const unsigned t_order = loc_order;
				loc_order = last_order + delay;
				last_order = t_order;
				numerrors = i + 1 - numerrors;
				last_disc = disc;
				delay = 1;
				continue;
			}
			// no more taps: loc[j + delay] -= last[j] * disc / last_disc for j <= last_order (last is not touched)
#pragma unroll
			for (int hh = 0; hh < 2; ++hh) {
				const unsigned k = (unsigned)lane + 64u * hh;
				if (k < 72u && k >= delay && k - delay <= last_order) loc[k] ^= gf_div(s, gf_mul(s, last[k - delay], disc), last_disc);
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			loc_order = (last_order + delay > loc_order) ? last_order + delay : loc_order;
			delay++;
		}
		const int order = (int)loc_order;
		__builtin_amdgcn_s_waitcnt(0);

		// error evaluator = locator * S mod x^30 (decode.c:149-161, polynomial.c:17-30): coefficient k on lane k
		if (lane < RS_PARITY) {
			uint8_t acc = 0;
			for (int i = 0; i <= order && i <= lane; ++i) acc ^= gf_mul(s, loc[i], synd[lane - i]);
			s.evalr[wv][lane] = acc;
		}
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_s_waitcnt(0);

		// Chien over every field element (decode.c:122-145); element 0 evaluates to loc[0] = 1, never a root
		int nroots = 0;
		uint8_t myroots[4];
		bool isroot[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			int e = lane + 64 * r;
			bool root = false;
			// order >= 30 would index past libcorrect's element_exp rows (undefined there); treated as "not a root"
			if (e != 0 && order < RS_PARITY) root = gf_eval(s, loc, order, (uint8_t)e) == 0;
			isroot[r] = root;
			myroots[r] = (uint8_t)e;
			nroots += __popcll(__ballot(root));
		}
		if (nroots != order) {
			ok = 0;   // decode.c:354-358: too many errors
		} else {
			// Forney (decode.c:165-196) + locations (decode.c:198-222) + fix-up (decode.c:366-369), one root per lane-slot
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				if (!isroot[r]) continue;
				const uint8_t e = myroots[r];
				// formal derivative (polynomial.c:74-87): der[i] = (i+1 odd) ? loc[i+1] : 0, order-1
				uint8_t num = gf_eval(s, s.evalr[wv], RS_PARITY - 1, e);
				uint8_t den = 0;
				{
					unsigned le = s.log[e] % 255u, acc = 0;
					for (int i = 0; i <= order - 1; ++i) {
						uint8_t c = ((i + 1) & 1) ? loc[i + 1] : 0;
						if (c) den ^= s.exp[(unsigned)s.log[c] + acc];
						acc += le; if (acc >= 255u) acc -= 255u;
					}
				}
				const uint8_t val = gf_div(s, num, den);                 // field_pow(root, fcr-1 = 0) == 1
				const uint8_t X = gf_div(s, 1, e);                        // error location = log(1/root); log(1) aliases to j = 0 -> 0
				const unsigned location = (X == 1) ? 0u : (unsigned)s.log[X];
				if (location < (unsigned)RS_BLOCK) enc[RS_BLOCK - 1 - location] ^= val;   // >= 155: lands in the zero padding, not emitted
			}
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_s_waitcnt(0);
		}
	}

	// the 125 message bytes of block b are bytes [125*(b%5), +125) of chunk first_chunk + b/5 (aligned_stream.h:62-85)
	uint8_t* dst = chunks + (size_t)f * FRAME_BYTES + (size_t)(first_chunk + b / 5) * CHUNK + (size_t)(b % 5) * RS_DATA;
	for (int k = lane; k < RS_DATA; k += 64) dst[k] = enc[k];
	if (lane == 0) rs_ok[(size_t)f * ALL_BLOCKS + ok_offset + b] = (uint8_t)ok;
}

// ------------------------------------------------------------------------------------------------ per-frame glue
// FountainMetadata.h:16-92
__device__ __forceinline__ uint32_t md_id(const uint8_t h[6]) { return (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16) | ((uint32_t)h[3] << 24); }
__device__ __forceinline__ unsigned md_file_size(const uint8_t h[6]) { return (unsigned)h[3] | ((unsigned)h[2] << 8) | ((unsigned)h[1] << 16) | (((unsigned)h[0] & 0x80u) << 17); }
__device__ __forceinline__ void md_increment(uint8_t h[6], unsigned radioactive)
{
	unsigned next = ((unsigned)h[5] | ((unsigned)h[4] << 8)) + 1;
	if (next == radioactive) next += 1;
	h[4] = (uint8_t)((next >> 8) & 0xFF);
	h[5] = (uint8_t)(next & 0xFF);
}

struct FrameState {           // aligned_stream state carried from the symbol to the colour pass
	uint32_t offset;          // aligned_stream::_offset
	uint32_t bad;             // aligned_stream::_badChunk
	uint32_t mask;            // chunks delivered so far (aligned_stream::_totalCount == 625 * popcount(mask))
	uint32_t pad;
};

// aligned_stream.h:39-119 driven one 125-byte RS block at a time (reed_solomon_stream.h:62-74,109-114). A bad LAST block of
// a chunk leaves _badChunk set, so the NEXT chunk is the one that gets dropped -- kept, it is what the reference does.
// `hdr`/`radio` mirror CimbReader::update_metadata (CimbReader.cpp:269-280) when track_md is set.
__device__ __forceinline__ void aligner_block(FrameState& st, int block_no, int ok, const uint8_t* frame_chunks, bool track_md,
                                              uint8_t hdr[6], unsigned& radio)
{
	const int chunk_index = block_no / 5;
	if (!ok) { st.bad = 1; st.offset = (st.offset + RS_DATA) % CHUNK; return; }
	if (RS_DATA + st.offset >= (unsigned)CHUNK) {
		const bool delivered = !st.bad;
		if (st.bad) { st.bad = 0; st.offset = 0; }
		else { st.mask |= 1u << chunk_index; st.offset = 0; }
		if (track_md) {
			if (!delivered && md_id(hdr) == 0) return;                       // update_metadata(nullptr, 0)
			if (md_id(hdr) == 0) { for (int k = 0; k < 6; ++k) hdr[k] = frame_chunks[(size_t)chunk_index * CHUNK + k]; }
			if (radio == 0) { unsigned fs = md_file_size(hdr); radio = (fs % CHUNK == 0) ? 0xFFFFFFFFu : fs / CHUNK; }
			md_increment(hdr, radio);
		}
		return;
	}
	st.offset += RS_DATA;
}

// Cell.h:30-62 mean_rgb_continuous(skip=false) over a 6x6 block whose top-left pixel is (x, y): uint16 sums / 36
__device__ __forceinline__ void mean6x6(const uint8_t* __restrict__ frame, int x, int y, uint32_t out[3])
{
	uint32_t r = 0, g = 0, b = 0;
#pragma unroll
	for (int i = 0; i < 6; ++i) {
		const uint8_t* p = frame + ((size_t)(y + i) * IMG + x) * 3;
#pragma unroll
		for (int j = 0; j < 6; ++j) { r += p[3 * j]; g += p[3 * j + 1]; b += p[3 * j + 2]; }
	}
	out[0] = (r & 0xFFFFu) / 36u; out[1] = (g & 0xFFFFu) / 36u; out[2] = (b & 0xFFFFu) / 36u;
}

// ---- float arithmetic that must match the CPU restatement operation for operation: no contraction, IEEE div/sqrt
#pragma clang fp contract(off)

// CimbReader.cpp:55-86 calculateWhite (dark): three 4x4 anchor-centre means (cv::mean -> double), max, floor (1,1,1).
// sums[a*3+c] = the integer sum of channel c over anchor a's 16 pixels; sum/16 is exact in double and in float.
__device__ __forceinline__ void white_from_sums(const uint32_t* sums, float white[3])
{
	white[0] = white[1] = white[2] = 1.0f;
	for (int a = 0; a < 3; ++a)
		for (int c = 0; c < 3; ++c) { float v = (float)((double)sums[a * 3 + c] / 16.0); if (v > white[c]) white[c] = v; }
}

// [assumed-OpenCV] lapack.cpp JacobiSVDImpl_<float>, n = 3 rows of length m = R (<= 5); same operation order as
// oracle/cimbar_oracle.c jacobi_svd_f32 (hypot spelled sqrt(p*p+beta*beta) in both)
__device__ void jacobi_svd3(float* At, int m, float* Wout, float* Vt)
{
	const int n = 3;
	const double minval = FLT_MIN;
	const float eps = FLT_EPSILON * 2;
	double W[3];
	const int max_iter = m > 30 ? m : 30;
	for (int i = 0; i < n; ++i) {
		double sd = 0;
		for (int k = 0; k < m; ++k) { float t = At[i * m + k]; sd += (double)t * t; }
		W[i] = sd;
		for (int k = 0; k < n; ++k) Vt[i * n + k] = 0;
		Vt[i * n + i] = 1;
	}
	for (int iter = 0; iter < max_iter; ++iter) {
		bool changed = false;
		for (int i = 0; i < n - 1; ++i)
			for (int j = i + 1; j < n; ++j) {
				float *Ai = At + i * m, *Aj = At + j * m;
				double a = W[i], p = 0, b = W[j];
				for (int k = 0; k < m; ++k) p += (double)Ai[k] * Aj[k];
				if (fabs(p) <= eps * sqrt(a * b)) continue;
				p *= 2;
				double beta = a - b, gamma = sqrt(p * p + beta * beta);
				float c, s;
				if (beta < 0) {
					double delta = (gamma - beta) * 0.5;
					s = (float)sqrt(delta / gamma);
					c = (float)(p / (gamma * s * 2));
				} else {
					c = (float)sqrt((gamma + beta) / (gamma * 2));
					s = (float)(p / (gamma * c * 2));
				}
				a = b = 0;
				for (int k = 0; k < m; ++k) {
					float t0 = c * Ai[k] + s * Aj[k];
					float t1 = -s * Ai[k] + c * Aj[k];
					Ai[k] = t0; Aj[k] = t1;
					a += (double)t0 * t0; b += (double)t1 * t1;
				}
				W[i] = a; W[j] = b;
				changed = true;
				float *Vi = Vt + i * n, *Vj = Vt + j * n;
				for (int k = 0; k < n; ++k) {
					float t0 = c * Vi[k] + s * Vj[k];
					float t1 = -s * Vi[k] + c * Vj[k];
					Vi[k] = t0; Vj[k] = t1;
				}
			}
		if (!changed) break;
	}
	for (int i = 0; i < n; ++i) {
		double sd = 0;
		for (int k = 0; k < m; ++k) { float t = At[i * m + k]; sd += (double)t * t; }
		W[i] = sqrt(sd);
	}
	for (int i = 0; i < n - 1; ++i) {
		int j = i;
		for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
		if (i != j) {
			double tw = W[i]; W[i] = W[j]; W[j] = tw;
			for (int k = 0; k < m; ++k) { float t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
			for (int k = 0; k < n; ++k) { float t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
		}
	}
	for (int i = 0; i < n; ++i) Wout[i] = (float)W[i];
	for (int i = 0; i < n; ++i) {
		double sd = W[i];
		float s = (float)(sd > minval ? 1 / sd : 0.);
		for (int k = 0; k < m; ++k) At[i * m + k] *= s;
	}
}

// color_correction.h:26-39 get_moore_penrose_lsm: ccm = desired^T * pinv(actual^T) [assumed-OpenCV: SVD + SVBkSb + gemm]
__device__ void moore_penrose_lsm(const float* actual, const float* desired, int R, float ccm[9])
{
	float A[15], V[9], W[3];
	for (int i = 0; i < 3; ++i) for (int k = 0; k < R; ++k) A[i * R + k] = actual[k * 3 + i];
	jacobi_svd3(A, R, W, V);
	float z[15];
	for (int i = 0; i < R * 3; ++i) z[i] = 0;
	double threshold = 0;
	for (int i = 0; i < 3; ++i) threshold += W[i];
	threshold *= (double)(FLT_EPSILON * 2);
	for (int k = 0; k < 3; ++k) {
		double wi = W[k];
		if (fabs(wi) <= threshold) continue;
		wi = 1 / wi;
		double buffer[3];
		for (int j = 0; j < 3; ++j) buffer[j] = V[k * 3 + j] * wi;
		for (int i = 0; i < R; ++i) {
			float s = A[k * R + i];
			for (int j = 0; j < 3; ++j) z[i * 3 + j] = (float)(z[i * 3 + j] + s * buffer[j]);
		}
	}
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double sm = 0;
			for (int k = 0; k < R; ++k) sm += (double)desired[k * 3 + i] * (double)z[k * 3 + j];
			ccm[i * 3 + j] = (float)sm;
		}
}

// color_correction.h:11-24 get_adaptation_matrix<von_kries>(white, (255,255,255)) -- color_correction == 1
__device__ void von_kries_ccm(const float white[3], float out[9])
{
	const float T[9] = {0.4002400f, 0.7076000f, -0.0808100f, -0.2263000f, 1.1653200f, 0.0457000f, 0.0000000f, 0.0000000f, 0.9182200f};
	float m1[3], m2[3], d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ti[9], tmp[9];
	for (int i = 0; i < 3; ++i) {
		float s = 0; for (int k = 0; k < 3; ++k) s += T[i * 3 + k] * white[k]; m1[i] = s;
		float q = 0; for (int k = 0; k < 3; ++k) q += T[i * 3 + k] * 255.0f; m2[i] = q;
	}
	for (int i = 0; i < 3; ++i) d[i * 3 + i] = m2[i] / m1[i];
#define A_(i, j) T[(i) * 3 + (j)]
	float det = (float)(A_(0,0) * (A_(1,1) * A_(2,2) - A_(2,1) * A_(1,2)) - A_(0,1) * (A_(1,0) * A_(2,2) - A_(2,0) * A_(1,2)) +
	                    A_(0,2) * (A_(1,0) * A_(2,1) - A_(2,0) * A_(1,1)));
	det = 1 / det;
	ti[0] = (A_(1,1) * A_(2,2) - A_(1,2) * A_(2,1)) * det; ti[1] = (A_(0,2) * A_(2,1) - A_(0,1) * A_(2,2)) * det;
	ti[2] = (A_(0,1) * A_(1,2) - A_(0,2) * A_(1,1)) * det; ti[3] = (A_(1,2) * A_(2,0) - A_(1,0) * A_(2,2)) * det;
	ti[4] = (A_(0,0) * A_(2,2) - A_(0,2) * A_(2,0)) * det; ti[5] = (A_(0,2) * A_(1,0) - A_(0,0) * A_(1,2)) * det;
	ti[6] = (A_(1,0) * A_(2,1) - A_(1,1) * A_(2,0)) * det; ti[7] = (A_(0,1) * A_(2,0) - A_(0,0) * A_(2,1)) * det;
	ti[8] = (A_(0,0) * A_(1,1) - A_(0,1) * A_(1,0)) * det;
#undef A_
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += ti[i * 3 + k] * d[k * 3 + j]; tmp[i * 3 + j] = s; }
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = 0; for (int k = 0; k < 3; ++k) s += tmp[i * 3 + k] * T[k * 3 + j]; out[i * 3 + j] = s; }
}

// CimbDecoder.cpp:168-200 get_best_color (+ :27-55, color_correction.h:64-68)
__device__ __forceinline__ uint32_t fix_single_color(float c, float adjust_up, float down)
{
	c -= down;
	c *= adjust_up;
	if (c > (245 - down)) c = 255;
	if (c < 0) c = 0;
	return (uint32_t)c & 0xFFu;
}
__device__ __forceinline__ uint32_t best_color(float r, float g, float b, const float* m, bool active)
{
	if (active) {
		float s0 = 0, s1 = 0, s2 = 0;
		s0 += m[0] * r; s0 += m[1] * g; s0 += m[2] * b;
		s1 += m[3] * r; s1 += m[4] * g; s1 += m[5] * b;
		s2 += m[6] * r; s2 += m[7] * g; s2 += m[8] * b;
		r = s0; g = s1; b = s2;
	}
	float mx = r; if (g > mx) mx = g; if (b > mx) mx = b; if (1.0f > mx) mx = 1.0f;
	float mn = r; if (g < mn) mn = g; if (b < mn) mn = b; if (48.0f < mn) mn = 48.0f;
	if (mn >= mx) mn = 0;
	// the reference computes 255.0/(max-min) in double and narrows to float (CimbDecoder.cpp:180). With both operands
	// exactly representable in binary32, double rounding through binary64 (53 >= 2*24+2 bits) is innocuous for division,
	// so the correctly rounded binary32 quotient is the same number -- and costs a third of the fp64 divide here.
	float adjust = __fdiv_rn(255.0f, mx - mn);
	int c0 = (int)fix_single_color(r, adjust, mn), c1 = (int)fix_single_color(g, adjust, mn), c2 = (int)fix_single_color(b, adjust, mn);
	int rel0 = c0 - c1, rel1 = c1 - c2, rel2 = c2 - c0;
	// CimbDecoder.cpp:186-198: the palette entry i minimising |rel - q_i|^2, q_i = (p0-p1, p1-p2, p2-p0), first minimum wins.
	// For the mode-B palette every q_i is a signed permutation of (255, -255, 0): |q_i|^2 is the same for all four, so the
	// minimum of the distance is the maximum of rel . q_i (first maximum wins) -- four differences instead of four
	// three-term squared distances:  q_0 = (-255, 255, 0), q_1 = (-255, 0, 255), q_2 = (0, 255, -255), q_3 = (255, -255, 0)
	// (the palette of c_palette -- {0,255,0},{0,255,255},{255,255,0},{255,0,255} -- is baked into these four differences)
	const int t0 = rel1 - rel0, t1 = rel2 - rel0, t2 = rel1 - rel2, t3 = rel0 - rel1;
	uint32_t best_fit = 0;
	int bt = t0;
	if (t1 > bt) { best_fit = 1; bt = t1; }
	if (t2 > bt) { best_fit = 2; bt = t2; }
	if (t3 > bt) { best_fit = 3; bt = t3; }
	return best_fit;
}
#pragma clang fp contract(fast)

// K4: one wavefront per frame, after the symbol RS pass. Runs the chunk bookkeeping for blocks 0..39, then
// CimbReader::init_ccm (CimbReader.cpp:169-267) for color_correction == 2, or the von Kries matrix for == 1.
// ccm_out[f] = {9 floats, valid}; valid == 0 means "keep whatever the thread had" (resolved in k_colors).
__global__ __launch_bounds__(64) void k_frame_mid(const uint8_t* __restrict__ rgb, const uint32_t* __restrict__ cellmean, Tables tb,
                                                  const uint8_t* __restrict__ chunks, const uint8_t* __restrict__ rs_ok,
                                                  int color_correction, FrameState* __restrict__ states, float* __restrict__ ccm_out, int f0,
                                                  int plain)
{
	const int f = f0 + blockIdx.x, lane = threadIdx.x;
	const uint8_t* frame = rgb + (size_t)f * FRAME_RGB;
	const uint8_t* fc = chunks + (size_t)f * FRAME_BYTES;
	__shared__ uint8_t s_hdr[4][6];      // predicted header of colour chunk c (CimbReader.cpp:188-227)
	__shared__ int s_have_hdr;
	__shared__ uint32_t s_cnt[4], s_r[4], s_g[4], s_b[4], s_first[4];

	// Everything this kernel reads from global memory is requested here, before anything waits: the kernel is one wavefront
	// per frame running a long serial chain (aligner, Jacobi SVD), so every dependent round trip to HBM shows in its latency.
	//   RS flags -> ballot, chunk headers -> LDS, the 96 header-predicted cells' means, the 48 anchor pixels of calculateWhite
	__shared__ uint8_t s_chunk_hdr[8 * CHUNK];   // only bytes [j*625, j*625+6) are filled / read (aligner_block's indexing)
	__shared__ uint32_t s_white[9];               // [anchor][channel] sums over the 4x4 anchor centres
	const uint8_t okb = lane < SYM_BLOCKS ? rs_ok[(size_t)f * ALL_BLOCKS + lane] : (uint8_t)0;
	const uint8_t hdrb = lane < 48 ? fc[(size_t)(lane / 6) * CHUNK + lane % 6] : (uint8_t)0;
	const bool need_cells = color_correction == 2, need_white = color_correction == 1 || color_correction == 2;
	uint32_t mv0 = 0, mv1 = 0, px[3] = {0, 0, 0};
	if (need_cells) {
		mv0 = cellmean[(size_t)f * GRID_CELLS + tb.ccm_grid[lane]];
		if (lane < 32) mv1 = cellmean[(size_t)f * GRID_CELLS + tb.ccm_grid[64 + lane]];
	}
	if (need_white && lane < 48) {
		// CimbReader.cpp:55-86 calculateWhite (dark): three 4x4 anchor-centre blocks; lane = anchor * 16 + row * 4 + col
		const int tl = ANCHOR - 2, far = IMG - ANCHOR - 2;
		const int a = lane >> 4, ax = a == 2 ? far : tl, ay = a == 1 ? far : tl;
		const uint8_t* p = frame + ((size_t)(ay + ((lane >> 2) & 3)) * IMG + (ax + (lane & 3))) * 3;
		px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
	}
	if (lane < 9) s_white[lane] = 0;
	const unsigned long long ok_bits = __ballot(okb != 0);
	if (lane < 48) s_chunk_hdr[(lane / 6) * CHUNK + lane % 6] = hdrb;
	__syncthreads();
	if (need_white && lane < 48) { atomicAdd(&s_white[(lane >> 4) * 3], px[0]); atomicAdd(&s_white[(lane >> 4) * 3 + 1], px[1]); atomicAdd(&s_white[(lane >> 4) * 3 + 2], px[2]); }
	if (lane == 0) {
		FrameState st = {0, 0, 0, 0};
		uint8_t hdr[6] = {0, 0, 0, 0, 0, 0};
		unsigned radio = 0;
		for (int b = 0; b < SYM_BLOCKS; ++b) aligner_block(st, b, (int)((ok_bits >> b) & 1ull), s_chunk_hdr, true, hdr, radio);
		states[f] = st;
		// Decoder::decode into a plain stream has no aligned_stream, so no fountain header ever reaches the reader (Decoder.h:163-169)
		s_have_hdr = !plain && md_id(hdr) != 0;
		for (int c = 0; c < 4; ++c) {
			for (int k = 0; k < 6; ++k) s_hdr[c][k] = hdr[k];
			md_increment(hdr, radio);
		}
	}
	if (lane < 4) { s_cnt[lane] = 0; s_r[lane] = 0; s_g[lane] = 0; s_b[lane] = 0; s_first[lane] = 0xFFFFFFFFu; }
	__syncthreads();

	float* out = ccm_out + (size_t)f * 10;
	if (color_correction == 1) {
		if (lane == 0) {
			float white[3], m[9];
			white_from_sums(s_white, white);
			von_kries_ccm(white, m);
			for (int k = 0; k < 9; ++k) out[k] = m[k];
			out[9] = 1.0f;
		}
		return;
	}
	if (color_correction != 2 || !s_have_hdr) {
		if (lane == 0) out[9] = 0.0f;
		return;
	}

	// 96 known-colour cells: colour-stream cells 3100*c + t, t < 24; expected colour = bits [2t, 2t+2) of header c. Their means
	// (undrifted grid position + 1, 6x6, CimbReader.cpp:216-217: exactly what K1 left in cellmean) were fetched at the top.
	for (int q = lane; q < 96; q += 64) {
		const int c = q / 24, t = q % 24;
		const uint32_t expected = ((uint32_t)s_hdr[c][t >> 2] >> (6 - 2 * (t & 3))) & 3u;
		const uint32_t mv = q < 64 ? mv0 : mv1;
		atomicAdd(&s_cnt[expected], 1u);
		atomicAdd(&s_r[expected], mv & 0xFFu);
		atomicAdd(&s_g[expected], (mv >> 8) & 0xFFu);
		atomicAdd(&s_b[expected], (mv >> 16) & 0xFFu);
		atomicMin(&s_first[expected], (uint32_t)q);
	}
	__syncthreads();

	if (lane == 0) {
		// rows in std::unordered_map iteration order = reverse order of first appearance (libstdc++)
		float actual[15], desired[15];
		int rows = 0;
		uint32_t used = 0;
		for (int pass = 0; pass < 4; ++pass) {
			int bestc = -1; uint32_t bestq = 0;
			for (int c = 0; c < 4; ++c)
				if (!(used & (1u << c)) && s_cnt[c] != 0 && (bestc < 0 || s_first[c] > bestq)) { bestc = c; bestq = s_first[c]; }
			if (bestc < 0) break;
			used |= 1u << bestc;
			actual[rows * 3] = (float)(s_r[bestc] / s_cnt[bestc]);
			actual[rows * 3 + 1] = (float)(s_g[bestc] / s_cnt[bestc]);
			actual[rows * 3 + 2] = (float)(s_b[bestc] / s_cnt[bestc]);
			desired[rows * 3] = (float)c_palette[bestc][0]; desired[rows * 3 + 1] = (float)c_palette[bestc][1]; desired[rows * 3 + 2] = (float)c_palette[bestc][2];
			++rows;
		}
		if (rows < 4) { out[9] = 0.0f; return; }
		float white[3], m[9];
		white_from_sums(s_white, white);
		actual[rows * 3] = white[0]; actual[rows * 3 + 1] = white[1]; actual[rows * 3 + 2] = white[2];
		desired[rows * 3] = desired[rows * 3 + 1] = desired[rows * 3 + 2] = 255.0f;
		++rows;
		moore_penrose_lsm(actual, desired, rows, m);
		for (int k = 0; k < 9; ++k) out[k] = m[k];
		out[9] = 1.0f;
	}
}

constexpr int K5_CELLS = 7;   // 7 * 256 = 1792 cells per workgroup -> 7 workgroups per frame
// K5: colour pass (Decoder.h:107-113; CimbReader.cpp:133-137; CimbDecoder.cpp:202-217). The matrix in force for frame
// f is the newest valid one among frames <= f of this batch, else the context's carried one (slot `carry`).
__global__ __launch_bounds__(256) void k_colors(const uint8_t* __restrict__ rgb, const uint32_t* __restrict__ cellmean, Tables tb,
                                                const float* __restrict__ ccm_frames,
                                                const float* __restrict__ carry, const uint32_t* __restrict__ flood_flag,
                                                const int8_t* __restrict__ drift, uint8_t* __restrict__ colors,
                                                float* __restrict__ ccm_used, int f0)
{
	const int f = f0 + blockIdx.y;
	__shared__ float s_m[10];
	if (threadIdx.x == 0) {
		int g = f;
		while (g >= 0 && ccm_frames[(size_t)g * 10 + 9] == 0.0f) --g;
		const float* src = g >= 0 ? ccm_frames + (size_t)g * 10 : carry;
		for (int k = 0; k < 10; ++k) s_m[k] = src[k];
		if (blockIdx.x == 0) for (int k = 0; k < 10; ++k) ccm_used[(size_t)f * 10 + k] = src[k];
	}
	__syncthreads();
	const bool flooded = flood_flag[f] != 0;
	const bool active = s_m[9] != 0.0f;
	// few, fat waves (the kernel is otherwise bound by the workgroup dispatch rate): K5_CELLS cells per lane
	uint32_t mv[K5_CELLS];
	if (!flooded) {   // all of a lane's loads go out before the first classifier runs
#pragma unroll
		for (int k = 0; k < K5_CELLS; ++k) {
			const int i = (blockIdx.x * K5_CELLS + k) * 256 + threadIdx.x;
			mv[k] = 0;
			if (i < NCELLS) mv[k] = cellmean[(size_t)f * GRID_CELLS + tb.cell_grid[i]];
		}
	}
#pragma unroll
	for (int k = 0; k < K5_CELLS; ++k) {
		const int i = (blockIdx.x * K5_CELLS + k) * 256 + threadIdx.x;
		if (i >= NCELLS) break;
		uint32_t col[3];
		if (flooded) {
			// the frame went through the exact flood pass: cells are read where their drift put them
			ushort2 xy = tb.cell_xy[i];
			const int x = (int)xy.x + drift[((size_t)f * NCELLS + i) * 2], y = (int)xy.y + drift[((size_t)f * NCELLS + i) * 2 + 1];
			mean6x6(rgb + (size_t)f * FRAME_RGB, x + 1, y + 1, col);
		} else {
			col[0] = mv[k] & 0xFFu; col[1] = (mv[k] >> 8) & 0xFFu; col[2] = (mv[k] >> 16) & 0xFFu;
		}
		colors[(size_t)f * NCELLS + i] = (uint8_t)best_color((float)col[0], (float)col[1], (float)col[2], s_m, active);
	}
}

// K7: chunk bookkeeping for the colour blocks, final mask, zero the slots of dropped chunks, per-frame good bytes
__global__ __launch_bounds__(64) void k_frame_end(const uint8_t* __restrict__ rs_ok, FrameState* __restrict__ states,
                                                  uint8_t* __restrict__ chunks, uint32_t* __restrict__ masks,
                                                  const float* __restrict__ ccm_used,
                                                  float* __restrict__ carry, int f0, int write_carry, int plain)
{
	const int f = f0 + blockIdx.x, lane = threadIdx.x;
	// the matrix carried into the next call = the one in force for the batch's last frame (CimbDecoder.cpp:69-85)
	if (write_carry && blockIdx.x == gridDim.x - 1 && lane < 10) carry[lane] = ccm_used[(size_t)f * 10 + lane];
	if (plain) {
		// Decoder::decode into a plain stream: every RS output is written where it falls, a failed block as 125 zero bytes
		// (reed_solomon_stream.h:62-74,96-107). masks[f] is not meaningful here; the per-block flags are in rs_ok.
		uint8_t* fb = chunks + (size_t)f * FRAME_BYTES;
		for (int b = 0; b < ALL_BLOCKS; ++b)
			if (!rs_ok[(size_t)f * ALL_BLOCKS + b])
				for (int k = lane; k < RS_DATA; k += 64) fb[(size_t)b * RS_DATA + k] = 0;
		if (lane == 0) masks[f] = 0;
		return;
	}
	__shared__ uint32_t s_mask;
	const unsigned long long ok_bits = __ballot(lane < COL_BLOCKS && rs_ok[(size_t)f * ALL_BLOCKS + SYM_BLOCKS + (lane < COL_BLOCKS ? lane : 0)] != 0);
	if (lane == 0) {
		FrameState st = states[f];
		uint8_t hdr[6] = {0, 0, 0, 0, 0, 0};
		unsigned radio = 0;
		for (int b = 0; b < COL_BLOCKS; ++b) aligner_block(st, SYM_BLOCKS + b, (int)((ok_bits >> b) & 1ull), nullptr, false, hdr, radio);
		states[f] = st;
		masks[f] = st.mask;
		s_mask = st.mask;
	}
	__syncthreads();
	const uint32_t mask = s_mask;
	uint8_t* fc = chunks + (size_t)f * FRAME_BYTES;
	for (int j = 0; j < CHUNKS; ++j)
		if (!(mask & (1u << j)))
			for (int k = lane; k < CHUNK; k += 64) fc[(size_t)j * CHUNK + k] = 0;
}

// ------------------------------------------------------------------------------------------------ encode half (frame synthesiser)
// E1: RS(155,125) encode, one block per wavefront (libcorrect encode.c:3-34: systematic, remainder of msg(x)*x^30 by the
// generator prod_{i=1..30}(x + alpha^i)). Lane j < 30 holds remainder coefficient j; every message byte is one LFSR step:
// fb = msg[i] ^ rem[29], rem[j] = rem[j-1] ^ fb * g[j]. The 155 code bytes are then split into per-cell symbol / colour values
// and scattered through the interleave map (Encoder.h:95-119: 4-bit symbols for blocks 0..39, 2-bit colours for 40..59).
__global__ __launch_bounds__(256) void k_rs_encode(const uint8_t* __restrict__ payload, Tables tb, const uint8_t* __restrict__ gen_log,
                                                   int nframes, uint8_t* __restrict__ symbols, uint8_t* __restrict__ colors)
{
	__shared__ uint8_t s_exp[512], s_log[256];
	__shared__ uint8_t s_code[4][160];
	for (int k = threadIdx.x; k < 512; k += 256) s_exp[k] = c_gf_exp[k];
	s_log[threadIdx.x] = c_gf_log[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int gb = blockIdx.x * 4 + wv;
	if (gb / ALL_BLOCKS >= nframes) return;
	const int f = gb / ALL_BLOCKS, b = gb % ALL_BLOCKS;
	const uint8_t* msg = payload + (size_t)f * FRAME_BYTES + (size_t)b * RS_DATA;
	uint8_t* code = s_code[wv];
	for (int k = lane; k < RS_DATA; k += 64) code[k] = msg[k];
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const uint32_t glog = lane < RS_PARITY ? gen_log[lane] : 0u;      // log of generator coefficient j (all non-zero)
	uint32_t rem = 0;
	for (int i = 0; i < RS_DATA; ++i) {
		const uint32_t top = (uint32_t)__builtin_amdgcn_readlane((int)rem, RS_PARITY - 1);
		const uint32_t fb = (uint32_t)code[i] ^ top;                      // uniform
		const uint32_t prev = from_left_lane(rem, 0u);                     // rem[j-1], 0 into lane 0
		const uint32_t prod = fb ? (uint32_t)s_exp[(uint32_t)s_log[fb] + glog] : 0u;
		rem = lane < RS_PARITY ? (prev ^ prod) : 0u;
	}
	if (lane < RS_PARITY) code[RS_DATA + (RS_PARITY - 1 - lane)] = (uint8_t)rem;   // parity, highest order first
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	for (int k = lane; k < RS_BLOCK; k += 64) {
		const uint32_t v = code[k];
		if (b < SYM_BLOCKS) {
			const int sidx = (RS_BLOCK * b + k) * 2;
			symbols[(size_t)f * NCELLS + tb.stream_cell[sidx]] = (uint8_t)(v >> 4);
			symbols[(size_t)f * NCELLS + tb.stream_cell[sidx + 1]] = (uint8_t)(v & 15u);
		} else {
			const int sidx = (RS_BLOCK * (b - SYM_BLOCKS) + k) * 4;
#pragma unroll
			for (int q = 0; q < 4; ++q) colors[(size_t)f * NCELLS + tb.stream_cell[sidx + q]] = (uint8_t)((v >> (6 - 2 * q)) & 3u);
		}
	}
}

// E2: render. Output-driven and write-coalesced: one wavefront per pixel row, lane l writes pixels [16l, 16l+16) as three
// 16-byte stores. A pixel inside a cell takes the palette colour where the tile has a foreground bit, black elsewhere
// (Common.cpp:150-171, dark mode); every other pixel comes from the template (CimbWriter.cpp:39-77).
__global__ __launch_bounds__(256) void k_render(const uint8_t* __restrict__ symbols, const uint8_t* __restrict__ colors,
                                                const uint8_t* __restrict__ tmpl, uint8_t* __restrict__ rgb)
{
	const int lane = threadIdx.x & 63;
	const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int f = blockIdx.y;
	const uint4* trow = reinterpret_cast<const uint4*>(tmpl + (size_t)y * IMG * 3 + lane * 48);
	uint4 t0 = trow[0], t1 = trow[1], t2 = trow[2];
	uint32_t d[12] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w};
	const int ry = y - OFFSET;
	if (ry >= 0 && ry < DIM * PITCH && ry % PITCH < 8) {
		const int row = ry / PITCH, dy = ry % PITCH;
#pragma unroll
		for (int p = 0; p < 16; ++p) {
			const int rx = lane * 16 + p - OFFSET;
			const int col = rx / PITCH, dx = rx - col * PITCH;
			if (rx >= 0 && col < DIM && dx < 8) {
				const int i = cell_index(row, col);
				if (i >= 0) {
					const uint32_t sym = symbols[(size_t)f * NCELLS + i], colr = colors[(size_t)f * NCELLS + i];
					const bool fg = (c_tile[sym] >> (63 - (dy * 8 + dx))) & 1ull;
					const uint32_t r = fg ? (uint32_t)c_palette[colr][0] : 0u, g = fg ? (uint32_t)c_palette[colr][1] : 0u,
					               bl = fg ? (uint32_t)c_palette[colr][2] : 0u;
					// bytes 3p, 3p+1, 3p+2 of the lane's 48
#pragma unroll
					for (int ch = 0; ch < 3; ++ch) {
						const int k = 3 * p + ch;
						const uint32_t v = ch == 0 ? r : (ch == 1 ? g : bl);
						d[k >> 2] = (d[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | (v << (8 * (k & 3)));
					}
				}
			}
		}
	}
	uint4* orow = reinterpret_cast<uint4*>(rgb + (size_t)f * FRAME_RGB + (size_t)y * IMG * 3 + lane * 48);
	orow[0] = make_uint4(d[0], d[1], d[2], d[3]);
	orow[1] = make_uint4(d[4], d[5], d[6], d[7]);
	orow[2] = make_uint4(d[8], d[9], d[10], d[11]);
}

// repack the internal word-oriented bitplane into CimbReader::_grayscale's byte layout (tap only)
__global__ void k_plane_bytes(const uint32_t* __restrict__ plane, uint8_t* __restrict__ out, size_t nwords)
{
	size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	uint32_t v = plane[w];
	out[4 * w] = (uint8_t)(v >> 24); out[4 * w + 1] = (uint8_t)(v >> 16); out[4 * w + 2] = (uint8_t)(v >> 8); out[4 * w + 3] = (uint8_t)v;
}

}  // namespace

// ================================================================================================ host side / C ABI
struct cimbar_hip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr;            // second half of a batch's tail kernels (see enqueue)
	hipEvent_t ev_k1 = nullptr, ev_join = nullptr, ev_mid[8] = {};
	int tail_split = 1, tail_parts = 2;
	std::string err;
	Tables tb{};
	// batch scratch (grown on demand)
	int cap = 0;
	int last_n = 0;
	uint8_t* d_rgb = nullptr;         // staging for host-resident input
	size_t d_rgb_cap = 0;
	uint32_t* d_plane = nullptr;
	uint32_t* d_cellmean = nullptr;
	uint8_t* d_symbols = nullptr;
	uint8_t* d_colors = nullptr;
	int8_t* d_drift = nullptr;
	uint32_t* d_flood = nullptr;
	uint8_t* d_rs_ok = nullptr;
	FrameState* d_states = nullptr;
	float* d_ccm_frames = nullptr;
	float* d_ccm_used = nullptr;
	// the pipelined entry point rotates through `pipe_depth` sets of the intermediates above (the members are always the set in
	// use by the newest batch; the others are parked here) and as many streams, so that several batches are in flight at once
	static constexpr int MAXP = 4;
	struct ScratchSet {
		int cap = 0;
		uint32_t* d_plane = nullptr; uint32_t* d_cellmean = nullptr; uint8_t* d_symbols = nullptr; uint8_t* d_colors = nullptr;
		int8_t* d_drift = nullptr; uint32_t* d_flood = nullptr; uint8_t* d_rs_ok = nullptr; FrameState* d_states = nullptr;
		float* d_ccm_frames = nullptr; float* d_ccm_used = nullptr;
	} parked[MAXP];
	int pipe_depth = 4;
	int pipe_set = 0;                 // which set the member pointers above currently are
	bool pipe_used[MAXP] = {};
	hipStream_t pstream[MAXP] = {};
	hipEvent_t ev_pk1[MAXP] = {}, ev_pdone[MAXP] = {};
	float* d_carry = nullptr;         // 10 floats
	uint8_t* d_template = nullptr;    // encode half: empty frame (background, anchors, guides)
	uint8_t* d_gen_log = nullptr;     // encode half: logs of the 30 low generator coefficients
	uint8_t* d_payload = nullptr;     // encode half: staging for host-resident payload
	size_t d_payload_cap = 0;
	uint8_t* d_chunks = nullptr;      // staging for host-resident output
	uint32_t* d_masks = nullptr;
	FloodScratch flood{};
	int flood_cap = 0;
	// timing
	bool timing = false;
	static constexpr int NSTAGE = 8;
	hipEvent_t ev[NSTAGE + 1] = {};
	float stage_ms[NSTAGE] = {};
};

namespace {

const char* const STAGE_NAMES[cimbar_hip_ctx::NSTAGE] = {"threshold", "symbols", "flood", "rs_symbols", "frame_mid", "colors", "rs_colors", "frame_end"};

#define HIPCHK(call)                                                                                      \
	do {                                                                                                  \
		hipError_t e__ = (call);                                                                          \
		if (e__ != hipSuccess) {                                                                          \
			ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                                \
			return CIMBAR_HIP_EHIP;                                                                       \
		}                                                                                                 \
	} while (0)

void host_cell_positions(std::vector<ushort2>& xy)
{
	// CellPositions.cpp:5-51 compute_linear for Conf8x8
	xy.resize(NCELLS);
	int n = 0;
	for (int i = 0; i < TOP_CELLS; ++i, ++n) xy[n] = make_ushort2((i % TOP_W) * PITCH + PITCH * MARKER + OFFSET, (i / TOP_W) * PITCH + OFFSET);
	for (int i = 0; i < MID_CELLS; ++i, ++n) xy[n] = make_ushort2((i % DIM) * PITCH + OFFSET, (i / DIM) * PITCH + MARKER * PITCH + OFFSET);
	for (int i = 0; i < TOP_CELLS; ++i, ++n) xy[n] = make_ushort2((i % TOP_W) * PITCH + PITCH * MARKER + OFFSET, (i / TOP_W) * PITCH + (DIM - MARKER) * PITCH + OFFSET);
}

// AdjacentCellFinder.cpp:16-105, literally (position look-ups included): the index arithmetic there has band-edge quirks
// (e.g. cells 494..499 have no "bottom", 11900..11905 no "top") that the flood order depends on.
struct AdjFinder {
	const std::vector<ushort2>& pos;
	static int in_row_with_margin(int index) { return (index < TOP_CELLS) ? 1 : (index < TOP_CELLS + MID_CELLS ? 0 : 1); }
	int right(int index) const
	{
		if (index < 0 || index >= NCELLS - 1) return -1;
		int next = index + 1;
		if (pos[next].x < pos[index].x) return -1;
		return next;
	}
	int left(int index) const
	{
		int next = index - 1;
		if (next < 0) return -1;
		if (pos[next].x > pos[index].x) return -1;
		return next;
	}
	int bottom(int index) const
	{
		if (index < 0 || index >= NCELLS) return -1;
		int inc = DIM;
		if (in_row_with_margin(index)) inc -= MARKER;
		int next = index + inc;
		if (in_row_with_margin(next)) next -= MARKER;
		if (next < 0 || next >= NCELLS) return -1;
		if (pos[next].x != pos[index].x) return -1;
		return next;
	}
	int top(int index) const
	{
		int inc = DIM;
		if (in_row_with_margin(index)) inc -= MARKER;
		int next = index - inc;
		if (in_row_with_margin(next)) next += MARKER;
		if (next < 0) return -1;
		if (pos[next].x != pos[index].x) return -1;
		return next;
	}
};

int build_tables(cimbar_hip_ctx* ctx)
{
	std::vector<ushort2> xy;
	host_cell_positions(xy);
	// Interleave.h:8-24 interleave_indices(12400, 155, 2): stream index -> linear cell index
	std::vector<uint16_t> sc;
	sc.reserve(NCELLS);
	const int part_size = NCELLS / 2;
	for (int part = 0; part < NCELLS; part += part_size)
		for (int chunk = 0; chunk < RS_BLOCK; ++chunk)
			for (int i = chunk; i < part_size; i += RS_BLOCK) sc.push_back((uint16_t)(i + part));
	// FloodDecodePositions::update's offer list per cell (FloodDecodePositions.cpp:85-129): adjacents, then both horizons
	std::vector<int16_t> cand((size_t)NCELLS * 12);
	AdjFinder finder{xy};
	for (int i = 0; i < NCELLS; ++i) {
		int16_t* c = &cand[(size_t)i * 12];
		const int rr = finder.right(i), ll = finder.left(i), dd = finder.bottom(i), uu = finder.top(i);
		c[0] = (int16_t)rr; c[1] = (int16_t)ll; c[2] = (int16_t)dd; c[3] = (int16_t)uu;
		for (int k = 4; k < 12; ++k) c[k] = -1;
		if (rr >= 0 && ll >= 0) {
			const int h0 = finder.right(rr), h2 = finder.left(ll);
			c[4] = (int16_t)h0; c[5] = (int16_t)(h0 >= 0 ? finder.right(h0) : -1);
			c[6] = (int16_t)h2; c[7] = (int16_t)(h2 >= 0 ? finder.left(h2) : -1);
		}
		if (uu >= 0 && dd >= 0) {
			const int v0 = finder.top(uu), v2 = finder.bottom(dd);
			c[8] = (int16_t)v0; c[9] = (int16_t)(v0 >= 0 ? finder.top(v0) : -1);
			c[10] = (int16_t)v2; c[11] = (int16_t)(v2 >= 0 ? finder.bottom(v2) : -1);
		}
		// k_flood evaluates the twelve offers of a step independently: that needs them to be distinct cells
		for (int a = 0; a < 12; ++a)
			for (int b = a + 1; b < 12; ++b)
				if (c[a] >= 0 && c[a] == c[b]) { ctx->err = "build_tables: duplicate cell in a flood offer list"; return CIMBAR_HIP_EINVAL; }
	}
	// GF(2^8) tables, libcorrect field.h:26-62 with primitive polynomial 0x187 (correct.h:159-160)
	uint8_t gexp[512], glog[256];
	unsigned element = 1;
	gexp[0] = 1; glog[0] = 0;
	for (unsigned i = 1; i < 512; ++i) {
		element *= 2;
		if (element > 255) element ^= 0x187;
		gexp[i] = (uint8_t)element;
		if (i < 256) glog[element] = (uint8_t)i;
	}
	HIPCHK(hipMalloc(&ctx->tb.cell_xy, sizeof(ushort2) * NCELLS));
	HIPCHK(hipMalloc(&ctx->tb.stream_cell, sizeof(uint16_t) * NCELLS));
	std::vector<uint16_t> cell_grid(NCELLS);
	for (int i = 0; i < NCELLS; ++i) cell_grid[i] = (uint16_t)((((int)xy[i].y - OFFSET) / PITCH) * DIM + ((int)xy[i].x - OFFSET) / PITCH);
	HIPCHK(hipMalloc(&ctx->tb.cell_grid, sizeof(uint16_t) * NCELLS));
	HIPCHK(hipMemcpy(ctx->tb.cell_grid, cell_grid.data(), sizeof(uint16_t) * NCELLS, hipMemcpyHostToDevice));
	std::vector<uint16_t> ccm_grid(96);
	for (int q = 0; q < 96; ++q) {
		const int cell = sc[3100 * (q / 24) + q % 24];
		ccm_grid[q] = (uint16_t)((((int)xy[cell].y - OFFSET) / PITCH) * DIM + ((int)xy[cell].x - OFFSET) / PITCH);
	}
	HIPCHK(hipMalloc(&ctx->tb.ccm_grid, sizeof(uint16_t) * 96));
	HIPCHK(hipMemcpy(ctx->tb.ccm_grid, ccm_grid.data(), sizeof(uint16_t) * 96, hipMemcpyHostToDevice));
	HIPCHK(hipMalloc(&ctx->tb.cand, sizeof(int16_t) * NCELLS * 12));
	HIPCHK(hipMemcpy(ctx->tb.cell_xy, xy.data(), sizeof(ushort2) * NCELLS, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(ctx->tb.stream_cell, sc.data(), sizeof(uint16_t) * NCELLS, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(ctx->tb.cand, cand.data(), sizeof(int16_t) * NCELLS * 12, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_gf_exp), gexp, 512));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_gf_log), glog, 256));
	// generator polynomial prod_{i=1..30} (x + alpha^i), low -> high (libcorrect reed-solomon.c:5-12, polynomial.c:205-240)
	uint8_t gen[RS_PARITY + 1] = {1};
	auto gmul = [&](uint8_t a, uint8_t b) -> uint8_t { return (!a || !b) ? 0 : gexp[(unsigned)glog[a] + glog[b]]; };
	for (int i = 0; i < RS_PARITY; ++i) {
		const uint8_t root = gexp[(i + 1) % 255];
		for (int j = i + 1; j >= 1; --j) gen[j] = gen[j - 1] ^ gmul(gen[j], root);
		gen[0] = gmul(gen[0], root);
	}
	uint8_t gen_log[RS_PARITY];
	for (int j = 0; j < RS_PARITY; ++j) gen_log[j] = glog[gen[j]];
	HIPCHK(hipMalloc(&ctx->d_gen_log, RS_PARITY));
	HIPCHK(hipMemcpy(ctx->d_gen_log, gen_log, RS_PARITY, hipMemcpyHostToDevice));
	return 0;
}

template <typename T>
hipError_t regrow(T*& p, size_t count)
{
	if (p) { hipError_t e = hipFree(p); p = nullptr; if (e != hipSuccess) return e; }
	return hipMalloc(&p, sizeof(T) * count);
}

int ensure_capacity(cimbar_hip_ctx* ctx, int n)
{
	if (n <= ctx->cap) return 0;
	size_t N = (size_t)n;
	HIPCHK(regrow(ctx->d_plane, N * PLANE_WORDS));
	HIPCHK(regrow(ctx->d_cellmean, N * GRID_CELLS));
	HIPCHK(regrow(ctx->d_symbols, N * NCELLS));
	HIPCHK(regrow(ctx->d_colors, N * NCELLS));
	HIPCHK(regrow(ctx->d_drift, N * NCELLS * 2));
	HIPCHK(regrow(ctx->d_flood, N));
	HIPCHK(regrow(ctx->d_rs_ok, N * ALL_BLOCKS));
	HIPCHK(regrow(ctx->d_states, N));
	HIPCHK(regrow(ctx->d_ccm_frames, N * 10));
	HIPCHK(regrow(ctx->d_ccm_used, N * 10));
	HIPCHK(regrow(ctx->d_chunks, N * FRAME_BYTES));
	HIPCHK(regrow(ctx->d_masks, N));
	if (!ctx->flood.heap) HIPCHK(regrow(ctx->flood.heap, (size_t)FLOOD_GRID * HEAP_CAP));   // one spill area per resident flood workgroup
	ctx->cap = n;
	return 0;
}

void destroy_ctx(cimbar_hip_ctx* ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	auto fr = [](void* p) { if (p) (void)hipFree(p); };
	fr(ctx->tb.cell_xy); fr(ctx->tb.stream_cell); fr(ctx->tb.cand); fr(ctx->tb.ccm_grid); fr(ctx->tb.cell_grid);
	fr(ctx->d_template); fr(ctx->d_gen_log); fr(ctx->d_payload); fr(ctx->d_rgb); fr(ctx->d_plane); fr(ctx->d_cellmean); fr(ctx->d_symbols); fr(ctx->d_colors); fr(ctx->d_drift); fr(ctx->d_flood);
	fr(ctx->d_rs_ok); fr(ctx->d_states); fr(ctx->d_ccm_frames); fr(ctx->d_ccm_used); fr(ctx->d_carry); fr(ctx->d_chunks);
	fr(ctx->d_masks); fr(ctx->flood.heap);
	for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
	for (int k = 2; k < cimbar_hip_ctx::MAXP; ++k) if (ctx->pstream[k]) (void)hipStreamDestroy(ctx->pstream[k]);
	for (hipEvent_t e : {ctx->ev_k1, ctx->ev_join}) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : ctx->ev_mid) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : ctx->ev_pk1) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : ctx->ev_pdone) if (e) (void)hipEventDestroy(e);
	for (auto& a : ctx->parked) {
		fr(a.d_plane); fr(a.d_cellmean); fr(a.d_symbols); fr(a.d_colors); fr(a.d_drift); fr(a.d_flood);
		fr(a.d_rs_ok); fr(a.d_states); fr(a.d_ccm_frames); fr(a.d_ccm_used);
	}
	delete ctx;
}

// pipelined batches in flight use the scratch sets and the tail stream: anything else that touches them on `st` waits first
int drain_pipeline_into(cimbar_hip_ctx* ctx, hipStream_t st)
{
	for (int k = 0; k < cimbar_hip_ctx::MAXP; ++k)
		if (ctx->pipe_used[k]) HIPCHK(hipStreamWaitEvent(st, ctx->ev_pdone[k], 0));   // (a completed event costs nothing to wait for)
	return 0;
}

// park the set in use, take the next one (each set is owned by exactly one of: the members, one parked[] slot)
void rotate_scratch_sets(cimbar_hip_ctx* ctx)
{
	auto exchange = [&](cimbar_hip_ctx::ScratchSet& a) {
		std::swap(ctx->cap, a.cap);
		std::swap(ctx->d_plane, a.d_plane); std::swap(ctx->d_cellmean, a.d_cellmean); std::swap(ctx->d_symbols, a.d_symbols);
		std::swap(ctx->d_colors, a.d_colors); std::swap(ctx->d_drift, a.d_drift); std::swap(ctx->d_flood, a.d_flood);
		std::swap(ctx->d_rs_ok, a.d_rs_ok); std::swap(ctx->d_states, a.d_states); std::swap(ctx->d_ccm_frames, a.d_ccm_frames);
		std::swap(ctx->d_ccm_used, a.d_ccm_used);
	};
	exchange(ctx->parked[ctx->pipe_set]);                       // members (set pipe_set) -> its slot; members now empty
	ctx->pipe_set = (ctx->pipe_set + 1) % ctx->pipe_depth;
	exchange(ctx->parked[ctx->pipe_set]);                       // slot of the next set -> members; that slot now empty
}

// enqueue the whole pipeline for n device-resident frames on stream `st`
int enqueue(cimbar_hip_ctx* ctx, hipStream_t st, const uint8_t* d_rgb, int n, int pre, int cc, uint8_t* d_chunks, uint32_t* d_masks, int plain = 0,
            bool pipe = false)
{
	const bool tm = ctx->timing && !pipe;
	int evi = 0;
	auto mark = [&]() -> hipError_t { return tm ? hipEventRecord(ctx->ev[evi++], st) : hipSuccess; };
	const dim3 cell_grid((NCELLS + 255) / 256, n);
	const int f0 = 0;   // kernels index frames as f0 + block index, so a caller may also run a sub-range of a resident batch

	HIPCHK(mark());
	{
		dim3 g(K1_STRIPS / 4, n);
		if (pre) hipLaunchKernelGGL((k_threshold<3, true>), g, dim3(256), 0, st, d_rgb, ctx->d_plane, ctx->d_cellmean, ctx->d_flood, f0);
		else hipLaunchKernelGGL((k_threshold<2, false>), g, dim3(256), 0, st, d_rgb, ctx->d_plane, ctx->d_cellmean, ctx->d_flood, f0);
	}
	HIPCHK(mark());
	// Everything after K1 is a chain of small kernels per frame (symbols -> flood -> RS -> header/CCM -> colours -> RS -> masks), some of
	// them latency-bound (k_frame_mid is one serial wavefront per frame). For a large batch the chain runs as two half-batches on two
	// streams, so one half's serial phases and launch gaps are covered by the other half's kernels. The only cross-half dependency
	// is the colour-correction carry: a frame with no matrix of its own takes the newest one of the frames before it, so the second
	// half's colour pass waits for the first half's k_frame_mid. Stage timing uses the single-stream order.
	const bool split = !tm && !pipe && ctx->tail_split && n >= 64 * ctx->tail_parts;
	auto tail = [&](hipStream_t s, int fa, int m, int part) -> hipError_t {
		// part 0: up to k_frame_mid, part 1: the rest
		if (part == 0) {
			hipLaunchKernelGGL(k_symbols, dim3(DIM / K2_BLOCK_ROWS, m), dim3(256), 0, s, ctx->d_plane, ctx->tb, ctx->d_symbols, ctx->d_flood, fa);
			if (hipError_t e = (s == st ? mark() : hipSuccess)) return e;
			{
				// the two halves of a split batch may run their flood kernels at the same time: each gets its own half of the spill areas
				const int areas = pipe ? FLOOD_GRID / ctx->pipe_depth : (split ? FLOOD_GRID / 2 : FLOOD_GRID);
				const int area0 = pipe ? ctx->pipe_set * areas : ((split && s != st) ? FLOOD_GRID / 2 : 0);
				hipLaunchKernelGGL(k_flood, dim3(m < areas ? m : areas), dim3(64), 0, s, ctx->d_plane, ctx->tb, ctx->flood, ctx->d_flood, ctx->d_symbols,
				                   ctx->d_drift, fa, m, area0);
			}
			if (hipError_t e = (s == st ? mark() : hipSuccess)) return e;
			hipLaunchKernelGGL((k_rs<4>), dim3((m * SYM_BLOCKS + 3) / 4), dim3(256), 0, s, ctx->d_symbols, ctx->tb, fa, m, 0, d_chunks, ctx->d_rs_ok, 0);
			if (hipError_t e = (s == st ? mark() : hipSuccess)) return e;
			hipLaunchKernelGGL(k_frame_mid, dim3(m), dim3(64), 0, s, d_rgb, ctx->d_cellmean, ctx->tb, d_chunks, ctx->d_rs_ok, cc, ctx->d_states, ctx->d_ccm_frames, fa, plain);
			return s == st ? mark() : hipSuccess;
		}
		hipLaunchKernelGGL(k_colors, dim3((NCELLS + 256 * K5_CELLS - 1) / (256 * K5_CELLS), m), dim3(256), 0, s, d_rgb, ctx->d_cellmean, ctx->tb, ctx->d_ccm_frames,
		                   ctx->d_carry, ctx->d_flood, ctx->d_drift, ctx->d_colors, ctx->d_ccm_used, fa);
		if (hipError_t e = (s == st ? mark() : hipSuccess)) return e;
		hipLaunchKernelGGL((k_rs<2>), dim3((m * COL_BLOCKS + 3) / 4), dim3(256), 0, s, ctx->d_colors, ctx->tb, fa, m, 8, d_chunks, ctx->d_rs_ok, SYM_BLOCKS);
		if (hipError_t e = (s == st ? mark() : hipSuccess)) return e;
		hipLaunchKernelGGL(k_frame_end, dim3(m), dim3(64), 0, s, ctx->d_rs_ok, ctx->d_states, d_chunks, d_masks, ctx->d_ccm_used, ctx->d_carry, fa,
		                   fa + m == n ? 1 : 0, plain);
		return s == st ? mark() : hipSuccess;
	};
	if (pipe) {
		// the whole batch runs on one of the context's pipeline streams (`st` here), the next batch on the next one: independent
		// queues, so K1 of one batch overlaps the short kernels of the others. The one thing that crosses over is the colour-
		// correction carry (and the order of the results): this batch's colour pass waits for the previous batch's end.
		const int set = ctx->pipe_set, prev = (set + ctx->pipe_depth - 1) % ctx->pipe_depth;
		HIPCHK(tail(st, f0, n, 0));
		if (ctx->pipe_used[prev]) HIPCHK(hipStreamWaitEvent(st, ctx->ev_pdone[prev], 0));
		HIPCHK(tail(st, f0, n, 1));
		HIPCHK(hipEventRecord(ctx->ev_pdone[set], st));
		ctx->pipe_used[set] = true;
	} else if (!split) {
		HIPCHK(tail(st, f0, n, 0));
		HIPCHK(tail(st, f0, n, 1));
	} else {
		// parts alternate between the caller's stream and stream2; pairs are issued together so that both streams always have work
		const int P = ctx->tail_parts;
		hipStream_t sb = ctx->stream2;
		auto lo = [&](int p) { return (int)((long long)n * p / P); };
		HIPCHK(hipEventRecord(ctx->ev_k1, st));
		HIPCHK(hipStreamWaitEvent(sb, ctx->ev_k1, 0));
		for (int p = 0; p < P; p += 2) {
			for (int q = p; q < p + 2; ++q) {
				hipStream_t sq = (q & 1) ? sb : st;
				HIPCHK(tail(sq, f0 + lo(q), lo(q + 1) - lo(q), 0));
				HIPCHK(hipEventRecord(ctx->ev_mid[q], sq));
			}
			for (int q = p; q < p + 2; ++q) {
				hipStream_t sq = (q & 1) ? sb : st;
				if (q > 0) HIPCHK(hipStreamWaitEvent(sq, ctx->ev_mid[q - 1], 0));   // every k_frame_mid of the frames before this part is done
				HIPCHK(tail(sq, f0 + lo(q), lo(q + 1) - lo(q), 1));
			}
		}
		HIPCHK(hipEventRecord(ctx->ev_join, sb));
		HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
	}
	HIPCHK(hipGetLastError());
	ctx->last_n = n;
	return 0;
}

}  // namespace

extern "C" {

int cimbar_hip_bufsize(void) { return FRAME_BYTES; }

int cimbar_hip_create(int device, int mode_val, cimbar_hip_ctx** out)
{
	if (!out) return CIMBAR_HIP_EINVAL;
	*out = nullptr;
	if (mode_val != 0 && mode_val != 68) return CIMBAR_HIP_EINVAL;   // Config.h:19-44: only Conf8x8 ("B") is implemented
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return CIMBAR_HIP_ENODEVICE;
	if (hipSetDevice(device) != hipSuccess) return CIMBAR_HIP_ENODEVICE;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CIMBAR_HIP_ENODEVICE;
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CIMBAR_HIP_ENODEVICE;   // kernels are built for gfx950 only

	cimbar_hip_ctx* ctx = new cimbar_hip_ctx();
	ctx->device = device;
	auto fail = [&](int code) { destroy_ctx(ctx); return code; };
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	if (const char* v = std::getenv("CIMBAR_HIP_PIPE_DEPTH")) { int k = std::atoi(v); if (k >= 2 && k <= cimbar_hip_ctx::MAXP) ctx->pipe_depth = k; }
	// As few streams as possible: the runtime multiplexes streams onto (by default) four hardware queues, and how the pipeline's
	// streams fall onto them matters (measured, 1024-frame batches: depth 4 over six streams 0.81 ms per batch, slower than depth 2;
	// depth 4 over these four streams 0.68 ms; raising GPU_MAX_HW_QUEUES to 8 made it worse again). So the pipeline reuses the two
	// streams the context has anyway and adds only what the depth needs beyond them.
	ctx->pstream[0] = ctx->stream;
	ctx->pstream[1] = ctx->stream2;
	for (int k = 2; k < ctx->pipe_depth; ++k)
		if (hipStreamCreateWithFlags(&ctx->pstream[k], hipStreamNonBlocking) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	for (hipEvent_t* e : {&ctx->ev_k1, &ctx->ev_join})
		if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	for (hipEvent_t& e : ctx->ev_mid) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	for (hipEvent_t& e : ctx->ev_pk1) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	for (hipEvent_t& e : ctx->ev_pdone) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	if (const char* v = std::getenv("CIMBAR_HIP_TAIL_SPLIT")) ctx->tail_split = std::atoi(v);
	if (const char* v = std::getenv("CIMBAR_HIP_TAIL_PARTS")) { int k = std::atoi(v); if (k >= 2 && k <= 8 && k % 2 == 0) ctx->tail_parts = k; }
	for (auto& e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	if (hipMalloc(&ctx->d_carry, sizeof(float) * 10) != hipSuccess) return fail(CIMBAR_HIP_ENOMEM);
	if (hipMemset(ctx->d_carry, 0, sizeof(float) * 10) != hipSuccess) return fail(CIMBAR_HIP_EHIP);
	if (build_tables(ctx) != 0) return fail(CIMBAR_HIP_EHIP);
	*out = ctx;
	return CIMBAR_HIP_OK;
}

void cimbar_hip_destroy(cimbar_hip_ctx* ctx) { destroy_ctx(ctx); }

const char* cimbar_hip_last_error(const cimbar_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int cimbar_hip_reset_ccm(cimbar_hip_ctx* ctx)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipDeviceSynchronize());   // pipelined batches in flight still carry the matrix forward
	HIPCHK(hipMemsetAsync(ctx->d_carry, 0, sizeof(float) * 10, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return 0;
}

int cimbar_hip_get_ccm(cimbar_hip_ctx* ctx, float out9[9])
{
	if (!ctx || !out9) return CIMBAR_HIP_EINVAL;
	float tmp[10];
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipDeviceSynchronize());   // whatever stream the last batch ran on
	HIPCHK(hipMemcpy(tmp, ctx->d_carry, sizeof tmp, hipMemcpyDeviceToHost));
	std::memcpy(out9, tmp, sizeof(float) * 9);
	return tmp[9] != 0.0f ? 1 : 0;
}

int64_t cimbar_hip_decode_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, int n, int rgb_mem, int should_preprocess,
                                int color_correction, uint8_t* chunks, uint32_t* masks, int out_mem, void* hip_stream)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	if (!rgb || !chunks || !masks || n <= 0) { ctx->err = "decode_batch: null buffer or n <= 0"; return CIMBAR_HIP_EINVAL; }
	if ((rgb_mem != CIMBAR_HIP_MEM_HOST && rgb_mem != CIMBAR_HIP_MEM_DEVICE) || (out_mem != CIMBAR_HIP_MEM_HOST && out_mem != CIMBAR_HIP_MEM_DEVICE)) {
		ctx->err = "decode_batch: rgb_mem / out_mem must be CIMBAR_HIP_MEM_HOST or CIMBAR_HIP_MEM_DEVICE";
		return CIMBAR_HIP_EINVAL;
	}
	HIPCHK(hipSetDevice(ctx->device));
	// NULL means what it means for any HIP launch -- the (legacy) null stream -- whenever a device buffer is involved, so the
	// work is ordered after whatever produced the frames there; the all-host path synchronises anyway and uses its own stream
	const bool any_device = rgb_mem == CIMBAR_HIP_MEM_DEVICE || out_mem == CIMBAR_HIP_MEM_DEVICE;
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : (any_device ? (hipStream_t)nullptr : ctx->stream);
	if (int r = drain_pipeline_into(ctx, st)) return r;
	if (int r = ensure_capacity(ctx, n)) return r;

	const uint8_t* d_rgb = rgb;
	if (rgb_mem == CIMBAR_HIP_MEM_HOST) {
		size_t need = (size_t)n * FRAME_RGB;
		if (need > ctx->d_rgb_cap) { HIPCHK(regrow(ctx->d_rgb, need)); ctx->d_rgb_cap = need; }
		HIPCHK(hipMemcpyAsync(ctx->d_rgb, rgb, need, hipMemcpyHostToDevice, st));
		d_rgb = ctx->d_rgb;
	}
	uint8_t* d_chunks = out_mem == CIMBAR_HIP_MEM_DEVICE ? chunks : ctx->d_chunks;
	uint32_t* d_masks = out_mem == CIMBAR_HIP_MEM_DEVICE ? masks : ctx->d_masks;

	if (int r = enqueue(ctx, st, d_rgb, n, should_preprocess, color_correction, d_chunks, d_masks)) return r;

	if (out_mem == CIMBAR_HIP_MEM_DEVICE) return 0;
	unsigned long long total = 0;
	HIPCHK(hipMemcpyAsync(chunks, d_chunks, (size_t)n * FRAME_BYTES, hipMemcpyDeviceToHost, st));
	HIPCHK(hipMemcpyAsync(masks, d_masks, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	if (ctx->timing)
		for (int k = 0; k < cimbar_hip_ctx::NSTAGE; ++k) HIPCHK(hipEventElapsedTime(&ctx->stage_ms[k], ctx->ev[k], ctx->ev[k + 1]));
	// aligned_stream::tellp() summed over the batch: 625 bytes per delivered chunk (aligned_stream.h:29-32)
	for (int f = 0; f < n; ++f) total += (unsigned long long)CHUNK * (unsigned)__builtin_popcount(masks[f] & 0xFFFu);
	return (int64_t)total;
}

int cimbar_hip_decode_batch_pipelined(cimbar_hip_ctx* ctx, const uint8_t* rgb, int n, int should_preprocess, int color_correction,
                                      uint8_t* chunks, uint32_t* masks, void* hip_stream)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	if (!rgb || !chunks || !masks || n <= 0) { ctx->err = "decode_batch_pipelined: null buffer or n <= 0"; return CIMBAR_HIP_EINVAL; }
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t st = (hipStream_t)hip_stream;
	rotate_scratch_sets(ctx);
	const int set = ctx->pipe_set;
	hipStream_t own = ctx->pstream[set];
	// the frames are whatever `hip_stream` has produced up to here
	HIPCHK(hipEventRecord(ctx->ev_pk1[set], st));
	HIPCHK(hipStreamWaitEvent(own, ctx->ev_pk1[set], 0));
	// (this set's intermediates belong to the batch issued pipe_depth calls ago on the same stream: stream order keeps them apart)
	if (int r = ensure_capacity(ctx, n)) return r;
	return enqueue(ctx, own, rgb, n, should_preprocess, color_correction, chunks, masks, 0, true);
}

int cimbar_hip_pipeline_depth(const cimbar_hip_ctx* ctx) { return ctx ? ctx->pipe_depth : CIMBAR_HIP_EINVAL; }

int cimbar_hip_pipeline_wait(cimbar_hip_ctx* ctx, void* hip_stream, int keep_newest)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t st = (hipStream_t)hip_stream;
	// batch j's colour pass waits for batch j-1's end, so the end of one batch implies the end of every earlier one
	if (keep_newest < 0) keep_newest = 0;
	if (keep_newest >= ctx->pipe_depth) return 0;
	const int target = (ctx->pipe_set + ctx->pipe_depth - keep_newest) % ctx->pipe_depth;
	if (ctx->pipe_used[target]) HIPCHK(hipStreamWaitEvent(st, ctx->ev_pdone[target], 0));
	return 0;
}

int64_t cimbar_hip_decode_plain_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, int n, int rgb_mem, int should_preprocess,
                                      int color_correction, uint8_t* bytes, uint8_t* block_ok, int out_mem, void* hip_stream)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	if (!rgb || !bytes || n <= 0) { ctx->err = "decode_plain_batch: null buffer or n <= 0"; return CIMBAR_HIP_EINVAL; }
	if ((rgb_mem != CIMBAR_HIP_MEM_HOST && rgb_mem != CIMBAR_HIP_MEM_DEVICE) || (out_mem != CIMBAR_HIP_MEM_HOST && out_mem != CIMBAR_HIP_MEM_DEVICE)) {
		ctx->err = "decode_plain_batch: rgb_mem / out_mem must be CIMBAR_HIP_MEM_HOST or CIMBAR_HIP_MEM_DEVICE";
		return CIMBAR_HIP_EINVAL;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const bool any_device = rgb_mem == CIMBAR_HIP_MEM_DEVICE || out_mem == CIMBAR_HIP_MEM_DEVICE;
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : (any_device ? (hipStream_t)nullptr : ctx->stream);
	if (int r = drain_pipeline_into(ctx, st)) return r;
	if (int r = ensure_capacity(ctx, n)) return r;
	const uint8_t* d_rgb = rgb;
	if (rgb_mem == CIMBAR_HIP_MEM_HOST) {
		size_t need = (size_t)n * FRAME_RGB;
		if (need > ctx->d_rgb_cap) { HIPCHK(regrow(ctx->d_rgb, need)); ctx->d_rgb_cap = need; }
		HIPCHK(hipMemcpyAsync(ctx->d_rgb, rgb, need, hipMemcpyHostToDevice, st));
		d_rgb = ctx->d_rgb;
	}
	uint8_t* d_bytes = out_mem == CIMBAR_HIP_MEM_DEVICE ? bytes : ctx->d_chunks;
	if (int r = enqueue(ctx, st, d_rgb, n, should_preprocess, color_correction, d_bytes, ctx->d_masks, 1)) return r;
	if (block_ok)
		HIPCHK(hipMemcpyAsync(block_ok, ctx->d_rs_ok, (size_t)n * ALL_BLOCKS, out_mem == CIMBAR_HIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
	if (out_mem == CIMBAR_HIP_MEM_DEVICE) return 0;
	HIPCHK(hipMemcpyAsync(bytes, d_bytes, (size_t)n * FRAME_BYTES, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	return (int64_t)n * FRAME_BYTES;   // what tellp() of the output stream advanced by: every block is written, good or zeroed
}

int cimbar_hip_decode_frame(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, size_t stride,
                            int should_preprocess, int color_correction, uint8_t* chunks, uint32_t* good_mask)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	if (!rgb || !chunks || !good_mask) { ctx->err = "decode_frame: null buffer"; return CIMBAR_HIP_EINVAL; }
	if (width != (unsigned)IMG || height != (unsigned)IMG) { ctx->err = "decode_frame: frame must be 1024x1024 RGB8"; return CIMBAR_HIP_EDIM; }
	if (stride == (size_t)IMG * 3 || stride == 0)
		return (int)cimbar_hip_decode_batch(ctx, rgb, 1, CIMBAR_HIP_MEM_HOST, should_preprocess, color_correction, chunks, good_mask, CIMBAR_HIP_MEM_HOST, nullptr);
	if (stride < (size_t)IMG * 3) { ctx->err = "decode_frame: stride < width*3"; return CIMBAR_HIP_EINVAL; }
	std::vector<uint8_t> packed(FRAME_RGB);
	for (int y = 0; y < IMG; ++y) std::memcpy(packed.data() + (size_t)y * IMG * 3, rgb + (size_t)y * stride, (size_t)IMG * 3);
	return (int)cimbar_hip_decode_batch(ctx, packed.data(), 1, CIMBAR_HIP_MEM_HOST, should_preprocess, color_correction, chunks, good_mask, CIMBAR_HIP_MEM_HOST, nullptr);
}

int cimbar_hip_set_template(cimbar_hip_ctx* ctx, const uint8_t* rgb_template, int mem)
{
	if (!ctx || !rgb_template) return CIMBAR_HIP_EINVAL;
	HIPCHK(hipSetDevice(ctx->device));
	if (!ctx->d_template) HIPCHK(hipMalloc(&ctx->d_template, FRAME_RGB));
	HIPCHK(hipMemcpy(ctx->d_template, rgb_template, FRAME_RGB, mem == CIMBAR_HIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
	return 0;
}

int cimbar_hip_encode_batch(cimbar_hip_ctx* ctx, const uint8_t* payload, int n, int payload_mem, uint8_t* rgb_out, int rgb_mem,
                            void* hip_stream)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	if (!payload || !rgb_out || n <= 0) { ctx->err = "encode_batch: null buffer or n <= 0"; return CIMBAR_HIP_EINVAL; }
	if (!ctx->d_template) { ctx->err = "encode_batch: call cimbar_hip_set_template first"; return CIMBAR_HIP_EINVAL; }
	if ((payload_mem != CIMBAR_HIP_MEM_HOST && payload_mem != CIMBAR_HIP_MEM_DEVICE) || (rgb_mem != CIMBAR_HIP_MEM_HOST && rgb_mem != CIMBAR_HIP_MEM_DEVICE)) {
		ctx->err = "encode_batch: payload_mem / rgb_mem must be CIMBAR_HIP_MEM_HOST or CIMBAR_HIP_MEM_DEVICE";
		return CIMBAR_HIP_EINVAL;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const bool any_device = payload_mem == CIMBAR_HIP_MEM_DEVICE || rgb_mem == CIMBAR_HIP_MEM_DEVICE;
	hipStream_t st = hip_stream ? (hipStream_t)hip_stream : (any_device ? (hipStream_t)nullptr : ctx->stream);
	if (int r = drain_pipeline_into(ctx, st)) return r;
	if (int r = ensure_capacity(ctx, n)) return r;
	const uint8_t* d_payload = payload;
	if (payload_mem == CIMBAR_HIP_MEM_HOST) {
		const size_t need = (size_t)n * FRAME_BYTES;
		if (need > ctx->d_payload_cap) { HIPCHK(regrow(ctx->d_payload, need)); ctx->d_payload_cap = need; }
		HIPCHK(hipMemcpyAsync(ctx->d_payload, payload, need, hipMemcpyHostToDevice, st));
		d_payload = ctx->d_payload;
	}
	uint8_t* d_out = rgb_out;
	if (rgb_mem == CIMBAR_HIP_MEM_HOST) {
		const size_t need = (size_t)n * FRAME_RGB;
		if (need > ctx->d_rgb_cap) { HIPCHK(regrow(ctx->d_rgb, need)); ctx->d_rgb_cap = need; }
		d_out = ctx->d_rgb;
	}
	hipLaunchKernelGGL(k_rs_encode, dim3((n * ALL_BLOCKS + 3) / 4), dim3(256), 0, st, d_payload, ctx->tb, ctx->d_gen_log, n, ctx->d_symbols, ctx->d_colors);
	hipLaunchKernelGGL(k_render, dim3(IMG / 4, n), dim3(256), 0, st, ctx->d_symbols, ctx->d_colors, ctx->d_template, d_out);
	HIPCHK(hipGetLastError());
	if (rgb_mem == CIMBAR_HIP_MEM_HOST) {
		HIPCHK(hipMemcpyAsync(rgb_out, d_out, (size_t)n * FRAME_RGB, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
	}
	return 0;
}

int64_t cimbar_hip_tap(cimbar_hip_ctx* ctx, int what, void* out, size_t out_bytes)
{
	if (!ctx || !out) return CIMBAR_HIP_EINVAL;
	if (ctx->last_n <= 0) { ctx->err = "tap: no batch has been decoded on this context yet"; return CIMBAR_HIP_EINVAL; }
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipDeviceSynchronize());
	const size_t n = (size_t)ctx->last_n;
	const void* src = nullptr;
	size_t bytes = 0;
	switch (what) {
		case CIMBAR_HIP_TAP_BITPLANE: {
			bytes = n * PLANE_WORDS * 4;
			if (out_bytes < bytes) { ctx->err = "tap: buffer too small"; return CIMBAR_HIP_EINVAL; }
			uint8_t* tmp = nullptr;
			HIPCHK(hipMalloc(&tmp, bytes));
			size_t nw = n * PLANE_WORDS;
			hipLaunchKernelGGL(k_plane_bytes, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_plane, tmp, nw);
			hipError_t e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, ctx->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			(void)hipFree(tmp);
			HIPCHK(e);
			return (int64_t)bytes;
		}
		case CIMBAR_HIP_TAP_SYMBOLS: src = ctx->d_symbols; bytes = n * NCELLS; break;
		case CIMBAR_HIP_TAP_COLORS: src = ctx->d_colors; bytes = n * NCELLS; break;
		case CIMBAR_HIP_TAP_DRIFT: src = ctx->d_drift; bytes = n * NCELLS * 2; break;
		case CIMBAR_HIP_TAP_RS_OK: src = ctx->d_rs_ok; bytes = n * ALL_BLOCKS; break;
		case CIMBAR_HIP_TAP_CCM: src = ctx->d_ccm_used; bytes = n * 10 * sizeof(float); break;
		case CIMBAR_HIP_TAP_FLOOD: {
			bytes = n;
			if (out_bytes < bytes) { ctx->err = "tap: buffer too small"; return CIMBAR_HIP_EINVAL; }
			std::vector<uint32_t> tmp(n);
			HIPCHK(hipMemcpy(tmp.data(), ctx->d_flood, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
			for (size_t k = 0; k < n; ++k) ((uint8_t*)out)[k] = tmp[k] ? 1 : 0;
			return (int64_t)bytes;
		}
#ifdef FLOOD_PROF
		case 100: {   // cycle counters of the flood kernel, 6 x u64 per frame: pop, decode, offers, pushes, #pops, #pushes
			bytes = n * 80;
			if (out_bytes < bytes) return CIMBAR_HIP_EINVAL;
			for (size_t k = 0; k < n; ++k) HIPCHK(hipMemcpy((uint8_t*)out + k * 80, ctx->flood.heap + k * HEAP_CAP, 80, hipMemcpyDeviceToHost));
			return (int64_t)bytes;
		}
#endif
		default: ctx->err = "tap: unknown selector"; return CIMBAR_HIP_EINVAL;
	}
	if (out_bytes < bytes) { ctx->err = "tap: buffer too small"; return CIMBAR_HIP_EINVAL; }
	HIPCHK(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
	if (what == CIMBAR_HIP_TAP_DRIFT) {
		// frames that took the parallel path never wrote their (all-zero) drift
		std::vector<uint32_t> fl(n);
		HIPCHK(hipMemcpy(fl.data(), ctx->d_flood, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
		for (size_t k = 0; k < n; ++k) if (!fl[k]) std::memset((uint8_t*)out + k * NCELLS * 2, 0, (size_t)NCELLS * 2);
	}
	return (int64_t)bytes;
}

int cimbar_hip_enable_timing(cimbar_hip_ctx* ctx, int on)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	ctx->timing = on != 0;
	return 0;
}

int cimbar_hip_stage_times(cimbar_hip_ctx* ctx, const char** names, float* ms, int max)
{
	if (!ctx) return CIMBAR_HIP_EINVAL;
	if (ctx->timing && ctx->last_n > 0) {
		// device-output batches do not synchronise inside decode_batch; resolve the events here
		HIPCHK(hipEventSynchronize(ctx->ev[cimbar_hip_ctx::NSTAGE]));
		for (int k = 0; k < cimbar_hip_ctx::NSTAGE; ++k) HIPCHK(hipEventElapsedTime(&ctx->stage_ms[k], ctx->ev[k], ctx->ev[k + 1]));
	}
	int k = 0;
	for (; k < cimbar_hip_ctx::NSTAGE && k < max; ++k) {
		if (names) names[k] = STAGE_NAMES[k];
		if (ms) ms[k] = ctx->stage_ms[k];
	}
	return k;
}

}  // extern "C"
