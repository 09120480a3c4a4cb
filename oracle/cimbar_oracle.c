/* oracle/cimbar_oracle.c -- plain-C restatement of libcimbar's mode-B frame decode path (one deskewed 1024x1024
 * RGB frame -> <=12 fountain chunks of 625 bytes), every function citing the reference file:line it follows
 * (paths relative to /root/reference/src).
 *
 * TEST INFRASTRUCTURE: the checker for libcimbar_amd's HIP path and bench.py's "port" CPU baseline. It is never
 * linked, imported or called by the product. See cimbar_oracle.h for the parity-pinning status.
 */
#include "cimbar_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ mode B constants
 * lib/cimb_translator/GridConf.h:121-142 (Conf8x8), Config.h:101-165 */
enum {
	IMG_W = CO_IMG_W, IMG_H = CO_IMG_H, CELL = 8, PITCH = 9, OFFSET = CO_OFFSET, DIM_X = CO_DIM_X, DIM_Y = CO_DIM_Y, MARKER = 6, /* lrint(54/9), GridConf.h:32-40 */
	TOP_W = DIM_X - 2 * MARKER,         /* 100 */
	TOP_CELLS = TOP_W * MARKER,         /* 600 */
	MID_CELLS = DIM_X * (DIM_Y - 2 * MARKER), /* 11200 | 7392 */
	NCELLS = CO_CELLS,
	SYM_BYTES = NCELLS * 4 / 8,         /* 6200 */
	COL_BYTES = NCELLS * 2 / 8,         /* 3100 */
	ANCHOR = 30
};

/* lib/cimb_translator/CimbDecoder.cpp:87-99 computes these from the embedded 8x8 tile PNGs (bitmaps.h) through
 * image_hash::average_hash (image_hash/average_hash.h:19-39). Values cross-checked against the reference build in
 * tests/test_oracle_vs_ref.py and against image_hash/test/averageHashTest.cpp:45-49 (tiles 0 and 1). */
static const uint64_t TILE_HASH[16] = {
	0xfffefcf8f0e0c080ULL, 0x80c0e0f0f8fcfeffULL, 0xff7f3f1f0f070301ULL, 0x0103070f1f3f7fffULL,
	0x181818ffff181818ULL, 0x66e7e70000e7e766ULL, 0x3c7ee7c3c3e77e3cULL, 0x18183c3c7e7effffULL,
	0xc0f0fcfffffcf0c0ULL, 0xfffcf00000f0fcffULL, 0xff3f0f00000f3fffULL, 0xe7e7e7e7c3c38181ULL,
	0x8181c3c3e7e7e7e7ULL, 0x0000c3e77e3c1800ULL, 0x0c1c387070381c0cULL, 0x1e1e38381c1c7878ULL
};

void co_geometry(int32_t o[10])
{
	o[0] = CO_MODE; o[1] = IMG_W; o[2] = IMG_H; o[3] = NCELLS; o[4] = CO_CHUNK; o[5] = CO_RS_BLOCK; o[6] = CO_RS_PARITY; o[7] = DIM_X; o[8] = DIM_Y; o[9] = OFFSET;
}

void co_tile_hashes(uint64_t out16[16]) { memcpy(out16, TILE_HASH, sizeof TILE_HASH); }

/* lib/cimb_translator/Common.cpp:21-31 getColor4 (colour_mode 1, Config.h:61-64); legacy modes use colour_mode 0 = getColor4_old (:45-54) */
#if CO_LEGACY && CO_COLOR_BITS == 3   /* getColor8_old, Common.cpp:71-84 */
static const uint8_t PALETTE[8][3] = {{0, 255, 255}, {127, 127, 255}, {255, 0, 255}, {255, 65, 65}, {255, 159, 0}, {255, 255, 0}, {255, 255, 255}, {0, 255, 0}};
#elif CO_LEGACY
static const uint8_t PALETTE[4][3] = {{0, 255, 255}, {255, 255, 0}, {255, 0, 255}, {0, 255, 0}};
#else
static const uint8_t PALETTE[4][3] = {{0, 255, 0}, {0, 255, 255}, {255, 255, 0}, {255, 0, 255}};
#endif

/* ------------------------------------------------------------------------------------------------ cell geometry */
/* lib/cimb_translator/CellPositions.cpp:5-51 compute_linear */
void co_cell_positions(int32_t* xy)
{
	int n = 0;
	for (int i = 0; i < TOP_CELLS; ++i, ++n) {
		xy[2 * n] = (i % TOP_W) * PITCH + PITCH * MARKER + OFFSET;
		xy[2 * n + 1] = (i / TOP_W) * PITCH + OFFSET;
	}
	for (int i = 0; i < MID_CELLS; ++i, ++n) {
		xy[2 * n] = (i % DIM_X) * PITCH + OFFSET;
		xy[2 * n + 1] = (i / DIM_X) * PITCH + MARKER * PITCH + OFFSET;
	}
	for (int i = 0; i < TOP_CELLS; ++i, ++n) {
		xy[2 * n] = (i % TOP_W) * PITCH + PITCH * MARKER + OFFSET;
		xy[2 * n + 1] = (i / TOP_W) * PITCH + (DIM_Y - MARKER) * PITCH + OFFSET;
	}
}

static int32_t g_pos[2 * NCELLS];
static int g_pos_init = 0;
static void ensure_pos(void) { if (!g_pos_init) { co_cell_positions(g_pos); g_pos_init = 1; } }

/* lib/cimb_translator/Interleave.h:8-36, size 12400, 155 chunks, 2 partitions (Config.h:157-165) */
static void interleave_indices(uint32_t* idx)
{
	unsigned n = 0, part_size = NCELLS / 2;
	for (unsigned part = 0; part < NCELLS; part += part_size)
		for (unsigned chunk = 0; chunk < CO_RS_BLOCK; ++chunk)
			for (unsigned i = chunk; i < part_size; i += CO_RS_BLOCK) idx[n++] = i + part;
}
void co_interleave_reverse(uint32_t* out)
{
	static uint32_t idx[NCELLS];
	interleave_indices(idx);
	for (unsigned src = 0; src < NCELLS; ++src) out[idx[src]] = src;
}

/* lib/cimb_translator/AdjacentCellFinder.cpp:16-105 */
static int in_row_with_margin(int index) { return (index < TOP_CELLS) ? 1 : (index < TOP_CELLS + MID_CELLS ? 0 : 1); }
static int adj_right(int index)
{
	if (index < 0 || index >= NCELLS - 1) return -1;
	int next = index + 1;
	if (g_pos[2 * next] < g_pos[2 * index]) return -1;
	return next;
}
static int adj_left(int index)
{
	int next = index - 1;
	if (next < 0) return -1;
	if (g_pos[2 * next] > g_pos[2 * index]) return -1;
	return next;
}
static int adj_bottom(int index)
{
	if (index < 0 || index >= NCELLS) return -1;
	int inc = DIM_X;
	if (in_row_with_margin(index)) inc -= MARKER;
	int next = index + inc;
	if (in_row_with_margin(next)) next -= MARKER;
	if (next < 0 || next >= NCELLS) return -1;
	if (g_pos[2 * next] != g_pos[2 * index]) return -1;
	return next;
}
static int adj_top(int index)
{
	int inc = DIM_X;
	if (in_row_with_margin(index)) inc -= MARKER;
	int next = index - inc;
	if (in_row_with_margin(next)) next += MARKER;
	if (next < 0) return -1;
	if (g_pos[2 * next] != g_pos[2 * index]) return -1;
	return next;
}
void co_adjacent(int index, int32_t out4[4])
{
	ensure_pos();
	out4[0] = adj_right(index); out4[1] = adj_left(index); out4[2] = adj_bottom(index); out4[3] = adj_top(index);
}

/* ------------------------------------------------------------------------------------------------ threshold + pack */
/* lib/cimb_translator/CimbReader.cpp:30-46 preprocessSymbolGrid; OpenCV arithmetic [assumed-OpenCV]:
 *   cvtColor(RGB2GRAY) 8u : (R*9798 + G*19235 + B*3735 + 2^14) >> 15
 *   filter2D(kernel CimbReader.cpp:17-21) : 4.5*c - n - s - w - e in float, BORDER_REFLECT_101, round-half-even, clamp
 *   adaptiveThreshold(MEAN_C, BINARY, block, C=0): mean = round(box_sum / block^2) with BORDER_REPLICATE (OpenCV's
 *     ((sum + divDelta) * divScale >> 23) equals (sum + block^2/2) / block^2 for every reachable sum), out = gray > mean
 * lib/bit_file/bitmatrix.h:14-46 mat_to_bitbuffer: 8 mask bytes -> 1 byte, MSB = leftmost */
void co_threshold_bitplane(const uint8_t* rgb, int w, int h, int preprocess, uint8_t* bitplane)
{
	size_t n = (size_t)w * h;
	uint8_t* gray = (uint8_t*)malloc(n);
	for (size_t i = 0; i < n; ++i)
		gray[i] = (uint8_t)((rgb[3 * i] * 9798 + rgb[3 * i + 1] * 19235 + rgb[3 * i + 2] * 3735 + (1 << 14)) >> 15);

	int block = 5;
	if (preprocess) {
		block = 7;
		uint8_t* sharp = (uint8_t*)malloc(n);
		for (int y = 0; y < h; ++y)
			for (int x = 0; x < w; ++x) {
				int yn = y - 1 < 0 ? 1 : y - 1, ys = y + 1 >= h ? h - 2 : y + 1;
				int xw = x - 1 < 0 ? 1 : x - 1, xe = x + 1 >= w ? w - 2 : x + 1;
				/* every term is a multiple of 0.5 below 2^11: exact in float, so tap order is immaterial */
				float acc = 4.5f * gray[(size_t)y * w + x] - gray[(size_t)yn * w + x] - gray[(size_t)y * w + xw] -
				            gray[(size_t)y * w + xe] - gray[(size_t)ys * w + x];
				long r = lrintf(acc);
				sharp[(size_t)y * w + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
			}
		free(gray);
		gray = sharp;
	}

	int rad = block / 2, area = block * block;
	uint16_t* rowsum = (uint16_t*)malloc(n * sizeof(uint16_t));
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			int acc = 0;
			for (int k = -rad; k <= rad; ++k) {
				int xx = x + k; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
				acc += gray[(size_t)y * w + xx];
			}
			rowsum[(size_t)y * w + x] = (uint16_t)acc;
		}
	memset(bitplane, 0, (n + 7) / 8);   /* (the reference packs a trailing partial byte differently, bitmatrix.h:36-46: never inside the grid) */
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			int acc = 0;
			for (int k = -rad; k <= rad; ++k) {
				int yy = y + k; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
				acc += rowsum[(size_t)yy * w + x];
			}
			int mean = (acc + area / 2) / area;
			if (gray[(size_t)y * w + x] > mean) {
				size_t bit = (size_t)y * w + x;
				bitplane[bit >> 3] |= (uint8_t)(0x80 >> (bit & 7));
			}
		}
	free(rowsum);
	free(gray);
}

/* lib/bit_file/bitbuffer.h:86-107 read(): `length` bits starting at bit `index`, MSB first */
static unsigned bits_read(const uint8_t* buf, size_t index, int length)
{
	unsigned res = 0;
	for (int k = 0; k < length; ++k) {
		size_t b = index + k;
		res = (res << 1) | ((buf[b >> 3] >> (7 - (b & 7))) & 1u);
	}
	return res;
}

/* ------------------------------------------------------------------------------------------------ symbol matching */
/* image_hash/average_hash.h:63-75 fuzzy_ahash(bitmatrix): ten 10-bit reads -> 100-bit window, row 0 on top;
 * image_hash/ahash_result.h:70-106 + bit_extractor.h:23-51: window id w = 8x8 block at column w%3, row w/3 */
static void window_hashes(const uint8_t* bitplane, int x0, int y0, uint64_t out9[9])
{
	unsigned rows[10];
	for (int i = 0; i < 10; ++i) rows[i] = bits_read(bitplane, (size_t)x0 + (size_t)(y0 + i) * IMG_W, 10);
	for (int wnd = 0; wnd < 9; ++wnd) {
		uint64_t hsh = 0;
		for (int k = 0; k < 8; ++k) hsh = (hsh << 8) | ((rows[wnd / 3 + k] >> (2 - wnd % 3)) & 0xFFu);
		out9[wnd] = hsh;
	}
}

/* lib/cimb_translator/CimbDecoder.cpp:101-132 get_best_symbol; visit order image_hash/ahash_result.h:26;
 * mode ALL (9 windows) iff cooldown == 0xFE (CimbDecoder.cpp:142-147), else FAST (5 windows) */
static unsigned best_symbol(const uint64_t h9[9], unsigned cooldown, unsigned* drift_offset, unsigned* best_distance)
{
	static const unsigned ORDER[9] = {4, 5, 7, 3, 1, 8, 0, 2, 6};
	unsigned nwin = (cooldown == 0xFE) ? 9 : 5;
	unsigned best_fit = 0;
	*drift_offset = 0;
	*best_distance = 1000;
	for (unsigned k = 0; k < nwin; ++k) {
		unsigned idx = ORDER[k];
		if (idx == cooldown && idx != 4) continue;
		for (unsigned t = 0; t < 16; ++t) {
			unsigned d = (unsigned)__builtin_popcountll(h9[idx] ^ TILE_HASH[t]); /* image_hash/hamming_distance.h:8-12 */
			if (d < *best_distance) {
				*best_distance = d; best_fit = t; *drift_offset = idx;
				if (d == 0) return best_fit;
			}
		}
	}
	return best_fit;
}

/* lib/cimb_translator/CellDrift.cpp:33-43 */
static unsigned calculate_cooldown(unsigned previous, unsigned idx)
{
	if (idx == 4) return 4;
	if (idx % 2 == 0) return 0xFF;
	if (((previous ^ idx) & 0xFF) == 6) return 0xFF;
	return idx;
}

/* libstdc++ std::priority_queue<decode_prio, vector, PrioCompare> (FloodDecodePositions.h:18-28,48): comp(a,b) = a.prio > b.prio.
 * bits/stl_heap.h __push_heap / __adjust_heap restated so that equal-priority ordering matches exactly. */
typedef struct { uint16_t idx; uint8_t prio; } hent;
typedef struct { hent* v; int n, cap; } heap_t;

static void heap_sift_up(hent* first, int hole, int top, hent value)
{
	int parent = (hole - 1) / 2;
	while (hole > top && first[parent].prio > value.prio) {
		first[hole] = first[parent];
		hole = parent;
		parent = (hole - 1) / 2;
	}
	first[hole] = value;
}
static void heap_push(heap_t* hp, hent e)
{
	if (hp->n == hp->cap) { hp->cap = hp->cap ? hp->cap * 2 : 1024; hp->v = (hent*)realloc(hp->v, (size_t)hp->cap * sizeof(hent)); }
	hp->v[hp->n++] = e;
	heap_sift_up(hp->v, hp->n - 1, 0, e);
}
static hent heap_pop(heap_t* hp)
{
	hent top = hp->v[0];
	if (hp->n > 1) {
		int len = hp->n - 1;           /* std::pop_heap: range [first, last-1) after moving the root out */
		hent value = hp->v[len];
		hp->v[len] = hp->v[0];
		int hole = 0, second = 0;
		while (second < (len - 1) / 2) {
			second = 2 * (second + 1);
			if (hp->v[second].prio > hp->v[second - 1].prio) second--;   /* comp(right, left) -> take left */
			hp->v[hole] = hp->v[second];
			hole = second;
		}
		if ((len & 1) == 0 && second == (len - 2) / 2) {
			second = 2 * (second + 1);
			hp->v[hole] = hp->v[second - 1];
			hole = second - 1;
		}
		heap_sift_up(hp->v, hole, 0, value);
	}
	hp->n--;
	return top;
}

typedef struct { int dx, dy; uint8_t prio, cooldown; } instr_t;

/* lib/cimb_translator/FloodDecodePositions.cpp:69-83 */
static void update_adjacents(const int adj[4], heap_t* hp, instr_t* ins, const uint8_t* remaining, int dx, int dy,
                             unsigned error_distance, uint8_t cooldown)
{
	for (int k = 0; k < 4; ++k) {
		int next = adj[k];
		if (next < 0 || !remaining[next]) continue;
		if (ins[next].prio <= error_distance) continue;
		ins[next].dx = dx; ins[next].dy = dy; ins[next].prio = (uint8_t)error_distance; ins[next].cooldown = cooldown;
		hent e = {(uint16_t)next, (uint8_t)error_distance};
		heap_push(hp, e);
	}
}

/* lib/cimb_translator/CimbReader.cpp:139-162 read() driven to completion, with FloodDecodePositions.cpp:17-134 */
/* statistics of the last co_symbol_pass on this thread (sizing of the GPU replay's LDS heap: tools/heap_peak.py): the largest number of live
 * priority-queue entries and the number of pops */
static __thread int g_heap_peak = 0, g_heap_pops = 0;
int co_last_heap_peak(void) { return g_heap_peak; }
int co_last_heap_pops(void) { return g_heap_pops; }

int co_symbol_pass(const uint8_t* bitplane, int32_t* visit, uint8_t* dist)
{
	ensure_pos();
	g_heap_peak = 0; g_heap_pops = 0;
	instr_t* ins = (instr_t*)malloc(sizeof(instr_t) * NCELLS);
	uint8_t* remaining = (uint8_t*)malloc(NCELLS);
	heap_t hp = {0, 0, 0};
	for (int i = 0; i < NCELLS; ++i) { remaining[i] = 1; ins[i].dx = ins[i].dy = 0; ins[i].prio = 0xFE; ins[i].cooldown = 0xFE; }

	/* seeds, FloodDecodePositions.cpp:27-41 */
	uint16_t small_row = TOP_W, last = NCELLS - 1, between = TOP_CELLS;
	hent seeds[8] = {{0, 0}, {(uint16_t)(small_row - 1), 0}, {last, 0}, {(uint16_t)(last - (small_row - 1)), 0},
	                 {between, 1}, {(uint16_t)(between + DIM_X - 1), 1}, {(uint16_t)(last - between), 1},
	                 {(uint16_t)(last - (between + DIM_X - 1)), 1}};
	for (int s = 0; s < 8; ++s) heap_push(&hp, seeds[s]);

	int count = 0;
	while (count < NCELLS && hp.n > 0) {
		if (hp.n > g_heap_peak) g_heap_peak = hp.n;
		hent e = heap_pop(&hp);
		++g_heap_pops;
		int i = e.idx;
		if (!remaining[i]) continue;
		remaining[i] = 0;

		int ddx = ins[i].dx, ddy = ins[i].dy;
		unsigned cooldown = ins[i].cooldown;
		int x = g_pos[2 * i] + ddx, y = g_pos[2 * i + 1] + ddy;

		uint64_t h9[9];
		window_hashes(bitplane, x - 1, y - 1, h9);
		unsigned drift_offset, error_distance;
		unsigned bits = best_symbol(h9, cooldown, &drift_offset, &error_distance);

		int bdx = (int)(drift_offset % 3) - 1, bdy = (int)(drift_offset / 3) - 1;   /* CellDrift.h:13-15 driftPairs */
		int ndx = ddx + bdx, ndy = ddy + bdy;                                     /* CellDrift.cpp:23-31, limit 7 */
		ndx = ndx > 7 ? 7 : (ndx < -7 ? -7 : ndx);
		ndy = ndy > 7 ? 7 : (ndy < -7 ? -7 : ndy);
		uint8_t ncool = (uint8_t)calculate_cooldown(cooldown, drift_offset);

		/* FloodDecodePositions.cpp:85-134 update() */
		int adj[4] = {adj_right(i), adj_left(i), adj_bottom(i), adj_top(i)};
		update_adjacents(adj, &hp, ins, remaining, ndx, ndy, error_distance, ncool);
		if (ins[i].prio < 3 && error_distance < 3 && ins[i].cooldown == 4 && ncool == 4) {
			if (adj[0] >= 0 && adj[1] >= 0) {
				int hz[4] = {-1, -1, -1, -1};
				hz[0] = adj_right(adj[0]);
				if (hz[0] >= 0) hz[1] = adj_right(hz[0]);
				hz[2] = adj_left(adj[1]);
				if (hz[2] >= 0) hz[3] = adj_left(hz[2]);
				update_adjacents(hz, &hp, ins, remaining, ndx, ndy, error_distance, ncool);
			}
			if (adj[3] >= 0 && adj[2] >= 0) {
				int vt[4] = {-1, -1, -1, -1};
				vt[0] = adj_top(adj[3]);
				if (vt[0] >= 0) vt[1] = adj_top(vt[0]);
				vt[2] = adj_bottom(adj[2]);
				if (vt[2] >= 0) vt[3] = adj_bottom(vt[2]);
				update_adjacents(vt, &hp, ins, remaining, ndx, ndy, error_distance, ncool);
			}
		}
		ins[i].prio = (uint8_t)error_distance;
		ins[i].cooldown = ncool;

		if (visit) { visit[4 * count] = i; visit[4 * count + 1] = x + bdx; visit[4 * count + 2] = y + bdy; visit[4 * count + 3] = (int32_t)bits; }
		if (dist) dist[count] = (uint8_t)error_distance;
		++count;
	}
	free(hp.v); free(remaining); free(ins);
	return count;
}

/* ------------------------------------------------------------------------------------------------ Reed-Solomon */
/* third_party_lib/libcorrect/include/correct/reed-solomon/field.h:26-62 field_create(0x187) */
static uint8_t gf_exp[512], gf_log[256];
static int gf_init_done = 0;
static void gf_init(void)
{
	if (gf_init_done) return;
	unsigned element = 1;
	gf_exp[0] = 1; gf_log[0] = 0;
	for (unsigned i = 1; i < 512; ++i) {
		element *= 2;
		if (element > 255) element ^= 0x187;
		gf_exp[i] = (uint8_t)element;
		if (i < 256) gf_log[element] = (uint8_t)i;
	}
	gf_init_done = 1;
}
static uint8_t gf_mul(uint8_t l, uint8_t r) { if (!l || !r) return 0; return gf_exp[(unsigned)gf_log[l] + gf_log[r]]; }   /* field.h:92-110 */
static uint8_t gf_div(uint8_t l, uint8_t r) { if (!l || !r) return 0; return gf_exp[255u + gf_log[l] - gf_log[r]]; }     /* field.h:112-131: x/0 = 0 */
static uint8_t gf_mul_log(uint8_t l, uint8_t r) { unsigned s = (unsigned)l + r; return (uint8_t)(s > 255 ? s - 255 : s); } /* field.h:133-146 */
static uint8_t gf_pow(uint8_t e, int p) { int m = (gf_log[e] * p) % 255; if (m < 0) m += 255; return gf_exp[m]; }          /* field.h:155-165 */

/* polynomial.c:160-171 polynomial_build_exp_lut */
static void build_exp_lut(uint8_t val, unsigned order, uint8_t* out)
{
	uint8_t ve = gf_log[1], vl = gf_log[val];
	for (unsigned i = 0; i <= order; ++i) {
		if (val == 0) out[i] = 0;
		else { out[i] = ve; ve = gf_mul_log(ve, vl); }
	}
}
/* polynomial.c:113-131 polynomial_eval_lut */
static uint8_t eval_lut(const uint8_t* coeff, unsigned order, const uint8_t* val_exp)
{
	if (val_exp[0] == 0) return coeff[0];
	uint8_t res = 0;
	for (unsigned i = 0; i <= order; ++i)
		if (coeff[i]) res ^= gf_exp[(unsigned)gf_log[coeff[i]] + val_exp[i]];
	return res;
}
/* polynomial.c:133-157 polynomial_eval_log_lut */
static uint8_t eval_log_lut(const uint8_t* coeff_log, unsigned order, const uint8_t* val_exp)
{
	if (val_exp[0] == 0) return coeff_log[0] == 0 ? 0 : gf_exp[coeff_log[0]];
	uint8_t res = 0;
	for (unsigned i = 0; i <= order; ++i)
		if (coeff_log[i]) res ^= gf_exp[(unsigned)coeff_log[i] + val_exp[i]];
	return res;
}

/* libcorrect/src/reed-solomon/decode.c:299-379 correct_reed_solomon_decode, block_length 255, fcr 1, root gap 1.
 * Literal, including: no syndrome re-check, no location<encoded_length check, x/0=0. */
int co_rs_decode(const uint8_t* enc, unsigned enc_len, unsigned parity, uint8_t* msg)
{
	gf_init();
	enum { MAXP = 64 };
	if (enc_len > 255 || parity >= MAXP || enc_len < parity) return -1;
	unsigned md = parity, msg_len = enc_len - md;
	uint8_t recv[256];
	memset(recv, 0, sizeof recv);
	for (unsigned i = 0; i < enc_len; ++i) recv[i] = enc[enc_len - (i + 1)];

	/* syndromes, decode.c:12-28: generator_roots[i] = exp[(gap*(i+fcr)) % 255] (reed-solomon.c:5-12) */
	uint8_t synd[MAXP], lut[256];
	int all_zero = 1;
	for (unsigned i = 0; i < md; ++i) {
		build_exp_lut(gf_exp[(1 * (i + 1)) % 255], 254, lut);
		synd[i] = eval_lut(recv, 254, lut);
		if (synd[i]) all_zero = 0;
	}
	if (all_zero) { for (unsigned i = 0; i < msg_len; ++i) msg[i] = recv[enc_len - (i + 1)]; return (int)msg_len; }

	/* Berlekamp-Massey, decode.c:32-118 */
	uint8_t loc[2 * MAXP + 2], last[2 * MAXP + 2];
	memset(loc, 0, sizeof loc); memset(last, 0, sizeof last);
	loc[0] = 1; last[0] = 1;
	unsigned loc_order = 0, last_order = 0, numerrors = 0, delay = 1;
	uint8_t last_disc = 1;
	for (unsigned i = 0; i < md; ++i) {
		uint8_t disc = synd[i];
		for (unsigned j = 1; j <= numerrors; ++j) disc ^= gf_mul(loc[j], synd[i - j]);
		if (!disc) { delay++; continue; }
		if (2 * numerrors <= i) {
			for (int j = (int)last_order; j >= 0; --j) last[j + delay] = gf_div(gf_mul(last[j], disc), last_disc);
			for (int j = (int)delay - 1; j >= 0; --j) last[j] = 0;
			for (unsigned j = 0; j <= last_order + delay; ++j) { uint8_t t = loc[j]; loc[j] ^= last[j]; last[j] = t; }
			unsigned t_order = loc_order;
			loc_order = last_order + delay;
			last_order = t_order;
			numerrors = i + 1 - numerrors;
			last_disc = disc;
			delay = 1;
			continue;
		}
		for (int j = (int)last_order; j >= 0; --j) loc[j + delay] ^= gf_div(gf_mul(last[j], disc), last_disc);
		loc_order = (last_order + delay > loc_order) ? last_order + delay : loc_order;
		delay++;
	}
	unsigned order = loc_order;

	/* Chien, decode.c:122-145 (+ :344-358) */
	uint8_t loc_log[2 * MAXP + 2], roots[2 * MAXP + 2];
	for (unsigned i = 0; i <= order; ++i) loc_log[i] = gf_log[loc[i]];
	unsigned nroots = 0;
	memset(roots, 0, sizeof roots);
	for (unsigned e = 0; e < 256; ++e) {
		build_exp_lut((uint8_t)e, md - 1, lut);
		/* element_exp rows hold only min_distance powers; a locator of order > md-1 cannot come out of BM here */
		if (!eval_log_lut(loc_log, order, lut)) { if (nroots < sizeof roots) roots[nroots] = (uint8_t)e; nroots++; }
	}
	if (nroots != order) return -1;

	/* locations, decode.c:198-222 */
	uint8_t locations[2 * MAXP + 2];
	memset(locations, 0, sizeof locations);
	for (unsigned i = 0; i < order; ++i) {
		if (roots[i] == 0) continue;
		uint8_t l = gf_div(1, roots[i]);
		for (unsigned j = 0; j < 256; ++j)
			if (gf_pow((uint8_t)j, 1) == l) { locations[i] = gf_log[j]; break; }
	}

	/* Forney, decode.c:165-196; evaluator = locator * S mod x^md (polynomial.c:17-30), derivative polynomial.c:74-87 */
	uint8_t evalr[MAXP], deriv[2 * MAXP + 2], vals[2 * MAXP + 2];
	memset(evalr, 0, sizeof evalr);
	for (unsigned i = 0; i <= order; ++i) {
		if (i > md - 1) continue;
		unsigned jl = (md - 1 > md - 1 - i) ? md - 1 - i : md - 1;
		for (unsigned j = 0; j <= jl; ++j) evalr[i + j] ^= gf_mul(loc[i], synd[j]);
	}
	memset(deriv, 0, sizeof deriv);
	for (unsigned i = 0; i + 1 <= order; ++i) deriv[i] = ((i + 1) % 2) ? loc[i + 1] : 0;
	memset(vals, 0, sizeof vals);
	for (unsigned i = 0; i < order; ++i) {
		if (roots[i] == 0) continue;
		build_exp_lut(roots[i], md - 1, lut);
		vals[i] = gf_mul(gf_pow(roots[i], 0), gf_div(eval_lut(evalr, md - 1, lut), eval_lut(deriv, order - 1, lut)));
	}
	for (unsigned i = 0; i < order; ++i) recv[locations[i]] ^= vals[i];
	for (unsigned i = 0; i < msg_len; ++i) msg[i] = recv[enc_len - (i + 1)];
	return (int)msg_len;
}

/* libcorrect/src/reed-solomon/encode.c:3-34 + polynomial.c:32-72 (polynomial_mod), generator reed-solomon.c:5-12 */
int co_rs_encode(const uint8_t* msg, unsigned msg_len, unsigned parity, uint8_t* enc)
{
	gf_init();
	if (parity >= 64 || msg_len > 255 - parity) return -1;
	uint8_t gen[65];
	memset(gen, 0, sizeof gen);
	gen[0] = 1;
	for (unsigned i = 0; i < parity; ++i) {          /* multiply by (x + alpha^(i+1)) */
		uint8_t root = gf_exp[(i + 1) % 255];
		for (int j = (int)i + 1; j >= 1; --j) gen[j] = gen[j - 1] ^ gf_mul(gen[j], root);
		gen[0] = gf_mul(gen[0], root);
	}
	uint8_t rem[65];
	memset(rem, 0, sizeof rem);
	for (unsigned i = 0; i < msg_len; ++i) {          /* LFSR division, high order first */
		uint8_t fb = msg[i] ^ rem[parity - 1];
		for (int j = (int)parity - 1; j >= 1; --j) rem[j] = rem[j - 1] ^ gf_mul(fb, gen[j]);
		rem[0] = gf_mul(fb, gen[0]);
	}
	memcpy(enc, msg, msg_len);
	for (unsigned i = 0; i < parity; ++i) enc[msg_len + i] = rem[parity - 1 - i];
	return (int)(msg_len + parity);
}

/* ------------------------------------------------------------------------------------------------ colour */
/* lib/cimb_translator/Cell.h:30-62 mean_rgb_continuous(skip=false): uint16 sums, integer divide */
static void cell_mean_rgb(const uint8_t* rgb, int x, int y, int cols, int rows, uint8_t out[3])
{
	uint16_t r = 0, g = 0, b = 0, count = 0;
	for (int i = 0; i < rows; ++i)
		for (int j = 0; j < cols; ++j, ++count) {
			const uint8_t* p = rgb + ((size_t)(y + i) * IMG_W + (x + j)) * 3;
			r += p[0]; g += p[1]; b += p[2];
		}
	if (!count) { out[0] = out[1] = out[2] = 0; return; }
	out[0] = (uint8_t)(r / count); out[1] = (uint8_t)(g / count); out[2] = (uint8_t)(b / count);
}

/* lib/cimb_translator/CimbDecoder.cpp:27-36 fix_single_color */
static uint8_t fix_single_color(float c, float adjust_up, float down)
{
	c -= down;
	c *= adjust_up;
	if (c > (245 - down)) c = 255;
	if (c < 0) c = 0;
	return (uint8_t)c;
}

/* lib/cimb_translator/CimbDecoder.cpp:168-200 get_best_color (+ :38-55 colour distance, chromatic_adaptation/color_correction.h:64-68) */
unsigned co_best_color(float r, float g, float b, const co_ccm* ccm)
{
	if (ccm && ccm->active) {
		const float* m = ccm->m;
		float s0 = 0, s1 = 0, s2 = 0;      /* Matx product: s = 0; s += m(i,k)*v(k) */
		s0 += m[0] * r; s0 += m[1] * g; s0 += m[2] * b;
		s1 += m[3] * r; s1 += m[4] * g; s1 += m[5] * b;
		s2 += m[6] * r; s2 += m[7] * g; s2 += m[8] * b;
		r = s0; g = s1; b = s2;
	}
	float mx = r; if (g > mx) mx = g; if (b > mx) mx = b; if (1.0f > mx) mx = 1.0f;
	float mn = r; if (g < mn) mn = g; if (b < mn) mn = b; if (48.0f < mn) mn = 48.0f;
	if (mn >= mx) mn = 0;
	float adjust = (float)(255.0 / (double)(mx - mn));
	int c0 = fix_single_color(r, adjust, mn), c1 = fix_single_color(g, adjust, mn), c2 = fix_single_color(b, adjust, mn);
	int rel[3] = {c0 - c1, c1 - c2, c2 - c0};
	unsigned best_fit = 0;
	float best_distance = 1000000;
	for (unsigned i = 0; i < (1u << CO_COLOR_BITS); ++i) {
		int p0 = PALETTE[i][0], p1 = PALETTE[i][1], p2 = PALETTE[i][2];
		int q[3] = {p0 - p1, p1 - p2, p2 - p0};
		unsigned d = (unsigned)((rel[0] - q[0]) * (rel[0] - q[0]) + (rel[1] - q[1]) * (rel[1] - q[1]) + (rel[2] - q[2]) * (rel[2] - q[2]));
		if (d < best_distance) { best_fit = i; best_distance = (float)d; }
	}
	return best_fit;
}

/* lib/cimb_translator/CimbReader.cpp:55-86 calculateWhite (dark): max over three 4x4 anchor-centre means, floor (1,1,1) */
static void calculate_white(const uint8_t* rgb, float white[3])
{
	int tl = ANCHOR - 2, right = IMG_W - ANCHOR - 2, bottom = IMG_H - ANCHOR - 2;
	int ax[3] = {tl, tl, right}, ay[3] = {tl, bottom, tl};
	white[0] = white[1] = white[2] = 1.0f;
	for (int a = 0; a < 3; ++a) {
		double s[3] = {0, 0, 0};
		for (int i = 0; i < 4; ++i)
			for (int j = 0; j < 4; ++j) {
				const uint8_t* p = rgb + ((size_t)(ay[a] + i) * IMG_W + (ax[a] + j)) * 3;
				s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
			}
		for (int c = 0; c < 3; ++c) { float v = (float)(s[c] / 16.0); if (v > white[c]) white[c] = v; }
	}
}

/* [assumed-OpenCV] lapack.cpp JacobiSVDImpl_<float> on the n rows (length m) of At; see oracle/cvshim for the notes */
static void jacobi_svd_f32(float* At, int astep, float* Wout, float* Vt, int vstep, int m, int n)
{
	const double minval = FLT_MIN;
	const float eps = FLT_EPSILON * 2;
	double W[8];
	int max_iter = m > 30 ? m : 30;
	for (int i = 0; i < n; ++i) {
		double sd = 0;
		for (int k = 0; k < m; ++k) { float t = At[i * astep + k]; sd += (double)t * t; }
		W[i] = sd;
		for (int k = 0; k < n; ++k) Vt[i * vstep + k] = 0;
		Vt[i * vstep + i] = 1;
	}
	for (int iter = 0; iter < max_iter; ++iter) {
		int changed = 0;
		for (int i = 0; i < n - 1; ++i)
			for (int j = i + 1; j < n; ++j) {
				float *Ai = At + i * astep, *Aj = At + j * astep;
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
				changed = 1;
				float *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
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
		for (int k = 0; k < m; ++k) { float t = At[i * astep + k]; sd += (double)t * t; }
		W[i] = sqrt(sd);
	}
	for (int i = 0; i < n - 1; ++i) {
		int j = i;
		for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
		if (i != j) {
			double tw = W[i]; W[i] = W[j]; W[j] = tw;
			for (int k = 0; k < m; ++k) { float t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t; }
			for (int k = 0; k < n; ++k) { float t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t; }
		}
	}
	for (int i = 0; i < n; ++i) Wout[i] = (float)W[i];
	for (int i = 0; i < n; ++i) {
		double sd = W[i];
		float s = (float)(sd > minval ? 1 / sd : 0.);
		for (int k = 0; k < m; ++k) At[i * astep + k] *= s;
	}
}

/* chromatic_adaptation/color_correction.h:26-39 get_moore_penrose_lsm(actual Rx3, desired Rx3), R = 5:
 * ccm = desired^T * pinv(actual^T); pinv through cv::invert(DECOMP_SVD) = Jacobi SVD + SVBkSb [assumed-OpenCV] */
static void moore_penrose_lsm(const float* actual, const float* desired, int R, float ccm[9])
{
	/* y = actual^T is 3 x R (m=3 < n=R): OpenCV runs the Jacobi on y's 3 rows of length R */
	float A[3 * 8], V[9], W[3];
	for (int i = 0; i < 3; ++i) for (int k = 0; k < R; ++k) A[i * R + k] = actual[k * 3 + i];
	jacobi_svd_f32(A, R, W, V, 3, R, 3);
	/* u(r,k) = V[k][r] (3x3), vt(k,c) = A[k][c] (3xR); z (R x 3) = sum_k vt_k^T (u_k / w_k) */
	float z[8 * 3];
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
	/* ccm = desired^T (3xR) * z (Rx3): double accumulator, cast to float */
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double s = 0;
			for (int k = 0; k < R; ++k) s += (double)desired[k * 3 + i] * (double)z[k * 3 + j];
			ccm[i * 3 + j] = (float)s;
		}
}

/* chromatic_adaptation/color_correction.h:11-24 get_adaptation_matrix<von_kries>(actual, (255,255,255)) -- color_correction==1 */
static void von_kries_ccm(const float white[3], float out[9])
{
	static const float T[9] = {0.4002400f, 0.7076000f, -0.0808100f, -0.2263000f, 1.1653200f, 0.0457000f, 0.0000000f, 0.0000000f, 0.9182200f};
	float m1[3], m2[3], d[9] = {0}, ti[9], tmp[9];
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

/* exported for the known-answer tests of chromatic_adaptation/test/color_correctionTest.cpp (values printed by a real OpenCV) */
void co_moore_penrose_lsm(const float* actual, const float* desired, int rows, float out9[9]) { moore_penrose_lsm(actual, desired, rows, out9); }
void co_von_kries_ccm(const float white[3], float out9[9]) { von_kries_ccm(white, out9); }
/* color_correction.h:64-68 transform(): Matx33f * Vec3f, the same sums co_best_color applies */
void co_ccm_transform(const float m[9], float r, float g, float b, float out3[3])
{
	for (int i = 0; i < 3; ++i) { float s = 0; s += m[i * 3] * r; s += m[i * 3 + 1] * g; s += m[i * 3 + 2] * b; out3[i] = s; }
}

/* ------------------------------------------------------------------------------------------------ fountain header */
/* lib/fountain/FountainMetadata.h:16-92 */
static uint32_t md_id(const uint8_t h[6]) { uint32_t v; memcpy(&v, h, 4); return v; }
static unsigned md_file_size(const uint8_t h[6]) { return (unsigned)h[3] | ((unsigned)h[2] << 8) | ((unsigned)h[1] << 16) | (((unsigned)h[0] & 0x80u) << 17); }
static void md_increment(uint8_t h[6], unsigned radioactive)
{
	unsigned next = ((unsigned)h[5] | ((unsigned)h[4] << 8)) + 1;
	if (next == radioactive) next += 1;
	h[4] = (uint8_t)((next >> 8) & 0xFF);     /* update_block_id_internal takes a uint16_t */
	h[5] = (uint8_t)(next & 0xFF);
}

typedef struct { uint8_t header[6]; unsigned radioactive; } md_state;

/* lib/cimb_translator/CimbReader.cpp:269-280 update_metadata (+ :99-104 computeRadioactiveBlockId) */
static void update_metadata(md_state* st, const uint8_t* buff, unsigned len)
{
	if (len == 0 && md_id(st->header) == 0) return;
	if (md_id(st->header) == 0) { memset(st->header, 0, 6); memcpy(st->header, buff, len > 6 ? 6 : len); }
	if (st->radioactive == 0) {
		unsigned fs = md_file_size(st->header);
		st->radioactive = (fs % CO_CHUNK == 0) ? 0xFFFFFFFFu : fs / CO_CHUNK;
	}
	md_increment(st->header, st->radioactive);
}

/* lib/encoder/aligned_stream.h:8-131 driven by reed_solomon_stream.h:54-77,109-114: one RS block at a time.
 * Blocks are 125 (mode 67: 143) bytes and chunks 625 (429), so a chunk always completes on a block boundary; the literal state machine is
 * kept because a bad LAST block of a chunk leaves _badChunk set and makes the NEXT chunk the one that is dropped. */
typedef struct { unsigned offset; int bad; unsigned total; unsigned nblocks; uint8_t buf[CO_CHUNK]; } aligner_t;

static void aligner_block(aligner_t* al, int ok, const uint8_t* data125, md_state* md, uint8_t* chunks, uint32_t* mask)
{
	unsigned chunk_index = al->nblocks / (CO_CHUNK / CO_RS_DATA);   /* 5 | 3 blocks per chunk */
	al->nblocks++;
	if (!ok) {                                  /* mark_bad_chunk(125), aligned_stream.h:97-104 */
		al->bad = 1;
		al->offset = (al->offset + CO_RS_DATA) % CO_CHUNK;
		return;
	}
	if (CO_RS_DATA + al->offset >= CO_CHUNK) {   /* aligned_stream.h:62-85 */
		if (al->bad) {
			al->bad = 0; al->offset = 0;
			update_metadata(md, NULL, 0);
		} else {
			memcpy(al->buf + al->offset, data125, CO_RS_DATA);
			al->offset += CO_RS_DATA;
			memcpy(chunks + (size_t)chunk_index * CO_CHUNK, al->buf, CO_CHUNK);   /* flush(): stream.write + callback */
			*mask |= 1u << chunk_index;
			update_metadata(md, al->buf, CO_CHUNK);
			al->total += al->offset;
			al->offset = 0;
		}
		return;
	}
	memcpy(al->buf + al->offset, data125, CO_RS_DATA);
	al->offset += CO_RS_DATA;
}

/* ------------------------------------------------------------------------------------------------ whole frame */
static __thread uint8_t t_symbols[NCELLS], t_colors[NCELLS];
static __thread int32_t t_positions[2 * NCELLS];
const uint8_t* co_last_symbols(void) { return t_symbols; }
const uint8_t* co_last_colors(void) { return t_colors; }
const int32_t* co_last_positions(void) { return t_positions; }

/* lib/cimb_translator/CimbReader.cpp:169-267 init_ccm (color_correction == 2) */
static void init_ccm(const uint8_t* rgb, md_state* md, co_ccm* ccm)
{
	if (md_id(md->header) == 0) return;
	static uint32_t idx[NCELLS];
	static int idx_init = 0;
	if (!idx_init) { interleave_indices(idx); idx_init = 1; }

	/* std::unordered_map<uint16_t,...> (CimbReader.cpp:200): with libstdc++ the four keys land in distinct buckets and
	 * iteration runs newest-first, i.e. in REVERSE order of first appearance */
	unsigned cnt[4] = {0}, sr[4] = {0}, sg[4] = {0}, sb[4] = {0};
	int first_seen[4], nseen = 0;
	const unsigned end = COL_BYTES * 8 / 2, interval = (NCELLS * 6 / 8) * 8 / CO_CHUNKS_PER_FRAME / 2, header_len = 6 * 8 / 2;
	for (unsigned block = 0; block < end; block += interval) {
		for (unsigned s = block, i = 0; s < block + header_len; ++s, i += 2) {
			unsigned expected = bits_read(md->header, i, 2);
			int cell = (int)idx[s];
			uint8_t col[3];
			cell_mean_rgb(rgb, g_pos[2 * cell] + 1, g_pos[2 * cell + 1] + 1, CELL - 2, CELL - 2, col);
			if (cnt[expected] == 0) first_seen[nseen++] = (int)expected;
			cnt[expected] += 1; sr[expected] += col[0]; sg[expected] += col[1]; sb[expected] += col[2];
		}
		md_increment(md->header, md->radioactive);
	}
	float actual[5 * 3], desired[5 * 3];
	int rows = 0;
	for (int k = nseen - 1; k >= 0; --k) {
		int c = first_seen[k];
		actual[rows * 3] = (float)(sr[c] / cnt[c]); actual[rows * 3 + 1] = (float)(sg[c] / cnt[c]); actual[rows * 3 + 2] = (float)(sb[c] / cnt[c]);
		desired[rows * 3] = PALETTE[c][0]; desired[rows * 3 + 1] = PALETTE[c][1]; desired[rows * 3 + 2] = PALETTE[c][2];
		++rows;
	}
	if (rows < 4) return;
	float white[3];
	calculate_white(rgb, white);
	actual[rows * 3] = white[0]; actual[rows * 3 + 1] = white[1]; actual[rows * 3 + 2] = white[2];
	desired[rows * 3] = desired[rows * 3 + 1] = desired[rows * 3 + 2] = 255;
	++rows;
	moore_penrose_lsm(actual, desired, rows, ccm->m);
	ccm->active = 1;
}

/* lib/encoder/Decoder.h:60-118 do_decode. plain == 0: behind decode_fountain's aligned_stream (Decoder.h:171-189): `out` = 12 chunk slots,
 * *good_mask = delivered chunks, returns the good bytes. plain != 0: Decoder::decode (Decoder.h:163-169) into a plain stream: `out` = the 60
 * RS outputs back to back, a failed block as 125 zero bytes (reed_solomon_stream.h:62-74,96-107), block_ok[b] = 1 where libcorrect
 * succeeded, returns what the stream's tellp() would (7500); no fountain header ever reaches the reader, so color_correction == 2
 * keeps whatever matrix the thread already had (CimbReader.cpp:169-180). */
static int do_decode(const uint8_t* rgb, int w, int h, int preprocess, int color_correction, co_ccm* ccm,
                     int plain, uint8_t* outbuf, uint32_t* good_mask, uint8_t* block_ok)
{
	co_ccm local = {{0}, 0};
	if (!ccm) ccm = &local;
	memset(outbuf, 0, (size_t)CO_CHUNKS_PER_FRAME * CO_CHUNK);
	if (good_mask) *good_mask = 0;
	if (w < IMG_W || h < IMG_H) {
		/* CimbReader::_good == false (CimbReader.cpp:119): done() at once, no cell is read. Decoder::do_decode (Decoder.h:81-117) still flushes its
		 * zero-initialised symbol and colour buffers through Reed-Solomon; an all-zero block is a valid codeword (the colour pass writes every
		 * cell's bits at bit 0 of the colour stream, one byte error at most, corrected), so every block "decodes" to zeros, aligned_stream
		 * delivers all 12 chunks and the header stays id 0: the full byte count, chunks of zeros, CCM untouched. */
		if (good_mask) *good_mask = (1u << CO_CHUNKS_PER_FRAME) - 1u;
		if (block_ok) memset(block_ok, 1, SYM_BYTES / CO_RS_BLOCK + COL_BYTES / CO_RS_BLOCK);
		return CO_CHUNKS_PER_FRAME * CO_CHUNK;
	}
	ensure_pos();
	/* A larger image (CimbReader.cpp:112-117): the grid sits _gridPadding = min(cols - image_size_x, rows - image_size_y) / 2 pixels in, in x
	 * and in y, and every later position -- cells, drift, the anchor centres of calculateWhite -- is relative to that origin. The threshold
	 * runs over the WHOLE image (its borders are the large image's borders); a cell window never leaves the grid's image_size_x x image_size_y
	 * window (offset 8 - 1 - 7 >= 0 ... ), so the window's bits and pixels, cut out, are all the rest of the decode ever reads. */
	uint8_t* crop = NULL;
	uint8_t* full_plane = NULL;
	const int pad = ((w - IMG_W) < (h - IMG_H) ? (w - IMG_W) : (h - IMG_H)) / 2;
	if (w != IMG_W || h != IMG_H) {
		full_plane = (uint8_t*)malloc(((size_t)w * h + 7) / 8);
		co_threshold_bitplane(rgb, w, h, preprocess, full_plane);
		crop = (uint8_t*)malloc((size_t)IMG_W * IMG_H * 3);
		for (int y = 0; y < IMG_H; ++y) memcpy(crop + (size_t)y * IMG_W * 3, rgb + ((size_t)(y + pad) * w + pad) * 3, (size_t)IMG_W * 3);
		rgb = crop;
	}

	static uint32_t rev[NCELLS];
	static int rev_init = 0;
	if (!rev_init) { co_interleave_reverse(rev); rev_init = 1; }

	/* CimbReader ctor, CimbReader.cpp:107-126 */
	uint8_t* bitplane = (uint8_t*)malloc((size_t)IMG_W * IMG_H / 8);
	if (full_plane) {
		memset(bitplane, 0, (size_t)IMG_W * IMG_H / 8);
		for (int y = 0; y < IMG_H; ++y)
			for (int x = 0; x < IMG_W; ++x) {
				const size_t src = (size_t)(y + pad) * w + (x + pad), dst = (size_t)y * IMG_W + x;
				if (full_plane[src >> 3] & (0x80 >> (src & 7))) bitplane[dst >> 3] |= (uint8_t)(0x80 >> (dst & 7));
			}
		free(full_plane);
	} else co_threshold_bitplane(rgb, IMG_W, IMG_H, preprocess, bitplane);
	if (color_correction == 1) {
		float white[3];
		calculate_white(rgb, white);
		von_kries_ccm(white, ccm->m);
		ccm->active = 1;
	}

	/* symbol pass, Decoder.h:81-102 */
	int32_t* visit = (int32_t*)malloc(sizeof(int32_t) * 4 * NCELLS);
	co_symbol_pass(bitplane, visit, NULL);
	uint8_t symbuf[SYM_BYTES], colbuf[COL_BYTES];
	memset(symbuf, 0, sizeof symbuf); memset(colbuf, 0, sizeof colbuf);
	for (int k = 0; k < NCELLS; ++k) {
		int i = visit[4 * k];
		unsigned bits = (unsigned)visit[4 * k + 3];
		unsigned bitpos = rev[i] * 4;                                      /* bitbuffer.h:62-84 write(bits, pos, 4) */
		symbuf[bitpos >> 3] |= (uint8_t)(bits << (4 - (bitpos & 7)));
		t_symbols[i] = (uint8_t)bits;
		t_positions[2 * i] = visit[4 * k + 1]; t_positions[2 * i + 1] = visit[4 * k + 2];
	}
	free(visit); free(bitplane);

	aligner_t al; memset(&al, 0, sizeof al);
	md_state md; memset(&md, 0, sizeof md);
	uint8_t out[CO_RS_DATA];
	uint32_t dummy_mask = 0;
	int nblock = 0;
#if CO_LEGACY
	{
		/* Decoder::do_decode_coupled, Decoder.h:121-161: one bit stream of 6-bit cells -- the symbol's 4 bits written as a 6-bit value at
		 * 6 * stream index (its top 2 bits zero), then the colour's 2 bits over those top 2 -- and ONE Reed-Solomon pass over its 60 blocks. No
		 * fountain header reaches the reader before the colour pass, so init_ccm is not called: the colour classifier runs with whatever
		 * matrix the thread carries (or the von Kries one of color_correction == 1 from the constructor). */
		static uint8_t bb[NCELLS * CO_CELL_BITS / 8];
		memset(bb, 0, sizeof bb);
		for (int i = 0; i < NCELLS; ++i) {
			uint8_t col[3];
			cell_mean_rgb(rgb, t_positions[2 * i] + 1, t_positions[2 * i + 1] + 1, CELL - 2, CELL - 2, col);
			const unsigned cbits = co_best_color(col[0], col[1], col[2], ccm);
			t_colors[i] = (uint8_t)cbits;
			const unsigned field = (cbits << 4) | t_symbols[i], pos = rev[i] * CO_CELL_BITS;
			for (int k = 0; k < CO_CELL_BITS; ++k)
				if (field & ((1u << (CO_CELL_BITS - 1)) >> k)) bb[(pos + k) >> 3] |= (uint8_t)(0x80u >> ((pos + k) & 7));
		}
		for (int b = 0; b < (int)sizeof bb / CO_RS_BLOCK; ++b, ++nblock) {
			int r = co_rs_decode(bb + b * CO_RS_BLOCK, CO_RS_BLOCK, CO_RS_PARITY, out);
			if (plain) { if (r > 0) memcpy(outbuf + (size_t)nblock * CO_RS_DATA, out, CO_RS_DATA); if (block_ok) block_ok[nblock] = r > 0; }
			else aligner_block(&al, r > 0, out, &md, outbuf, good_mask ? good_mask : &dummy_mask);
		}
		free(crop);
		return plain ? nblock * CO_RS_DATA : (int)al.total;
	}
#endif
	for (int b = 0; b < SYM_BYTES / CO_RS_BLOCK; ++b, ++nblock) {          /* reed_solomon_stream.h:54-77 */
		int r = co_rs_decode(symbuf + b * CO_RS_BLOCK, CO_RS_BLOCK, CO_RS_PARITY, out);
		if (plain) { if (r > 0) memcpy(outbuf + (size_t)nblock * CO_RS_DATA, out, CO_RS_DATA); if (block_ok) block_ok[nblock] = r > 0; }
		else aligner_block(&al, r > 0, out, &md, outbuf, good_mask ? good_mask : &dummy_mask);
	}

	if (color_correction == 2) init_ccm(rgb, &md, ccm);                    /* Decoder.h:105 */

	/* colour pass, Decoder.h:107-117; CimbReader.cpp:133-137; CimbDecoder.cpp:202-217 */
	for (int i = 0; i < NCELLS; ++i) {
		uint8_t col[3];
		cell_mean_rgb(rgb, t_positions[2 * i] + 1, t_positions[2 * i + 1] + 1, CELL - 2, CELL - 2, col);
		unsigned bits = co_best_color(col[0], col[1], col[2], ccm);
		unsigned bitpos = rev[i] * 2;
		colbuf[bitpos >> 3] |= (uint8_t)(bits << (6 - (bitpos & 7)));
		t_colors[i] = (uint8_t)bits;
	}
	for (int b = 0; b < COL_BYTES / CO_RS_BLOCK; ++b, ++nblock) {
		int r = co_rs_decode(colbuf + b * CO_RS_BLOCK, CO_RS_BLOCK, CO_RS_PARITY, out);
		if (plain) { if (r > 0) memcpy(outbuf + (size_t)nblock * CO_RS_DATA, out, CO_RS_DATA); if (block_ok) block_ok[nblock] = r > 0; }
		else aligner_block(&al, r > 0, out, &md, outbuf, good_mask ? good_mask : &dummy_mask);
	}
	free(crop);
	return plain ? nblock * CO_RS_DATA : (int)al.total;
}

int co_decode_fountain(const uint8_t* rgb, int w, int h, int preprocess, int color_correction, co_ccm* ccm,
                       uint8_t* chunks, uint32_t* good_mask)
{
	return do_decode(rgb, w, h, preprocess, color_correction, ccm, 0, chunks, good_mask, NULL);
}

int co_decode_plain(const uint8_t* rgb, int w, int h, int preprocess, int color_correction, co_ccm* ccm,
                    uint8_t* bytes, uint8_t* block_ok)
{
	return do_decode(rgb, w, h, preprocess, color_correction, ccm, 1, bytes, NULL, block_ok);
}
