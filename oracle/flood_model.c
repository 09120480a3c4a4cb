/* oracle/flood_model.c -- CPU model of the "confluent flood" fast path of the HIP decoder (k_flood_wave in
 * libcimbar_amd/csrc/k2c_floodwave.hip.inc), TEST INFRASTRUCTURE like the rest of oracle/.
 *
 * The reference decodes the cells of a frame in the order of a std::priority_queue (FloodDecodePositions.cpp:49-134) and a
 * cell inherits drift / cooldown from whichever neighbour made the best offer before it popped. Seen abstractly that is
 * Prim's algorithm: "repeatedly decode SOME not-yet-decoded cell whose best offered priority is minimal", libstdc++'s heap
 * only decides WHICH of several tied cells goes first. This model runs the first `prefix` steps exactly (real heap, exact tie
 * order -- the garbage phase of a shifted frame is genuinely order dependent), then continues in "super-rounds": all cells
 * tied at the minimal priority p, and every cell that joins them at priority p, are decoded level by level as ONE batch,
 * while checking that no tie-break could have changed any cell's observable outcome:
 *   B1  a cell tied with others must not have two candidate inputs (equal priority, same super-round) with different outcomes
 *   B2  a member's own distance d < p is only allowed when the super-round is that single cell (else its cheaper offers could
 *       overtake a tied cell)
 *   B3  a member's offer must not be better than the input another member of the same super-round used
 *   B4  ... nor tie with it with a different outcome
 *   B5  more than two distinct candidate inputs for one cell (not tracked)
 *   B6  a "horizon" offer whose precondition depends on the tie-break must have no effect
 * If none fires, every schedule the reference's heap could have taken gives the same symbols and positions, so the batch
 * result IS the reference's. If one fires the caller falls back to the exact replay (co_symbol_pass here, k_flood on the GPU).
 * tests/test_flood_model.py checks exactly that claim against co_symbol_pass on distorted frames. */
#include "cimbar_oracle.c"

typedef struct { int sym, w, dist, ndx, ndy, ncool; } out_t;

static inline uint16_t pack_k(int dx, int dy, unsigned cool) { return (uint16_t)(((dx + 8) << 8) | ((dy + 8) << 4) | (cool == 0xFE ? 0 : (cool == 0xFF ? 2 : cool))); }
static inline int k_dx(uint16_t k) { return (int)(k >> 8) - 8; }
static inline int k_dy(uint16_t k) { return (int)((k >> 4) & 15) - 8; }
static inline unsigned k_cool(uint16_t k) { unsigned c = k & 15; return c == 0 ? 0xFE : (c == 2 ? 0xFF : c); }
#define K_DEFAULT ((uint16_t)((8 << 8) | (8 << 4) | 0))

static out_t decode_cell(const uint8_t* bitplane, int i, uint16_t k)
{
	out_t o;
	int ddx = k_dx(k), ddy = k_dy(k);
	unsigned cooldown = k_cool(k);
	int x = g_pos[2 * i] + ddx, y = g_pos[2 * i + 1] + ddy;
	uint64_t h9[9];
	window_hashes(bitplane, x - 1, y - 1, h9);
	unsigned w, d;
	o.sym = (int)best_symbol(h9, cooldown, &w, &d);
	o.w = (int)w; o.dist = (int)d;
	int ndx = ddx + (int)(w % 3) - 1, ndy = ddy + (int)(w / 3) - 1;
	o.ndx = ndx > 7 ? 7 : (ndx < -7 ? -7 : ndx);
	o.ndy = ndy > 7 ? 7 : (ndy < -7 ? -7 : ndy);
	o.ncool = (int)calculate_cooldown(cooldown, w);
	return o;
}
/* same observable result and same offers (the `far` precondition is compared separately) */
static int out_equal(const out_t* a, const out_t* b, uint16_t ka, uint16_t kb)
{
	return a->sym == b->sym && a->dist == b->dist && a->ndx == b->ndx && a->ndy == b->ndy && a->ncool == b->ncool &&
	       k_dx(ka) + a->w % 3 == k_dx(kb) + b->w % 3 && k_dy(ka) + a->w / 3 == k_dy(kb) + b->w / 3;
}

static int seed_prio(int i)
{
	const int last = NCELLS - 1;
	if (i == 0 || i == TOP_W - 1 || i == last || i == last - (TOP_W - 1)) return 0;
	if (i == TOP_CELLS || i == TOP_CELLS + DIM_X - 1 || i == last - TOP_CELLS || i == last - (TOP_CELLS + DIM_X - 1)) return 1;
	return 0xFE;
}

static void targets_of(int i, int t[12])
{
	for (int k = 0; k < 12; ++k) t[k] = -1;
	int rr = adj_right(i), ll = adj_left(i), dd = adj_bottom(i), uu = adj_top(i);
	t[0] = rr; t[1] = ll; t[2] = dd; t[3] = uu;
	if (rr >= 0 && ll >= 0) {
		t[4] = adj_right(rr); if (t[4] >= 0) t[5] = adj_right(t[4]);
		t[6] = adj_left(ll); if (t[6] >= 0) t[7] = adj_left(t[6]);
	}
	if (uu >= 0 && dd >= 0) {
		t[8] = adj_top(uu); if (t[8] >= 0) t[9] = adj_top(t[8]);
		t[10] = adj_bottom(dd); if (t[10] >= 0) t[11] = adj_bottom(t[10]);
	}
}

/* stats: [0] bail code (0 = certified), [1] super-rounds, [2] levels, [3] cells decoded by the prefix, [4] cells decoded in batches,
 * [5] cell at which it bailed, [6] equivalence checks that passed, [7] uncertain horizon offers checked */
int fm_flood(const uint8_t* bitplane, int prefix, uint8_t* sym_out, int32_t* pos_out, int32_t* stats)
{
	ensure_pos();
	static __thread uint8_t vis[NCELLS], b[NCELLS], sp[NCELLS], bpop[NCELLS], queued[NCELLS];
	static __thread uint16_t k[NCELLS], alt[NCELLS], kused[NCELLS];
	static __thread int rnd[NCELLS], mrnd[NCELLS];
	static __thread out_t outs[NCELLS];
	static __thread int queue[NCELLS], level[NCELLS];
	typedef struct { int t; uint16_t k; int d; } unc_t;
	static __thread unc_t unc[NCELLS];
	for (int s = 0; s < 8; ++s) stats[s] = 0;
	for (int i = 0; i < NCELLS; ++i) { vis[i] = 0; b[i] = 0xFE; sp[i] = (uint8_t)seed_prio(i); k[i] = K_DEFAULT; alt[i] = 0xFFFF; rnd[i] = 0; mrnd[i] = -1; queued[i] = 0; }
	int decoded = 0;

	/* ---- exact prefix: co_symbol_pass's loop (FloodDecodePositions.cpp + CimbReader.cpp:139-162), state kept in b / k */
	{
		heap_t hp = {0, 0, 0};
		uint16_t small_row = TOP_W, last = NCELLS - 1, between = TOP_CELLS;
		hent seeds[8] = {{0, 0}, {(uint16_t)(small_row - 1), 0}, {last, 0}, {(uint16_t)(last - (small_row - 1)), 0},
		                 {between, 1}, {(uint16_t)(between + DIM_X - 1), 1}, {(uint16_t)(last - between), 1},
		                 {(uint16_t)(last - (between + DIM_X - 1)), 1}};
		for (int s = 0; s < 8; ++s) heap_push(&hp, seeds[s]);
		while (decoded < prefix && decoded < NCELLS && hp.n > 0) {
			hent e = heap_pop(&hp);
			int i = e.idx;
			if (vis[i]) continue;
			vis[i] = 1;
			out_t o = decode_cell(bitplane, i, k[i]);
			sym_out[i] = (uint8_t)o.sym;
			pos_out[2 * i] = g_pos[2 * i] + k_dx(k[i]) + o.w % 3 - 1;
			pos_out[2 * i + 1] = g_pos[2 * i + 1] + k_dy(k[i]) + o.w / 3 - 1;
			int t[12];
			targets_of(i, t);
			const int far = b[i] < 3 && o.dist < 3 && k_cool(k[i]) == 4 && o.ncool == 4;
			const uint16_t nk = pack_k(o.ndx, o.ndy, (unsigned)o.ncool);
			for (int q = 0; q < (far ? 12 : 4); ++q) {
				int c = t[q];
				if (c < 0 || vis[c]) continue;
				if (b[c] <= o.dist) continue;
				b[c] = (uint8_t)o.dist; k[c] = nk;
				hent ne = {(uint16_t)c, (uint8_t)o.dist};
				heap_push(&hp, ne);
			}
			++decoded;
		}
		free(hp.v);
		stats[3] = decoded;
	}

	/* ---- super-rounds */
	int round = 0;
#define BAIL(code, cell) do { stats[0] = (code); stats[5] = (cell); return (code); } while (0)
	while (decoded < NCELLS) {
		++round;
		int p = 0xFE;
		for (int i = 0; i < NCELLS; ++i) if (!vis[i]) { int q = sp[i] < b[i] ? sp[i] : b[i]; if (q < p) p = q; }
		if (p == 0xFE) break;   /* unreachable cells stay undecoded, as in the reference */
		int nq = 0, nunc = 0, members = 0;
		for (int i = 0; i < NCELLS; ++i) if (!vis[i] && (sp[i] < b[i] ? sp[i] : b[i]) == p) { queue[nq++] = i; queued[i] = 1; }
		const int tied = nq;
		stats[1]++;
		int ended = 0;
		while (nq > 0 && !ended) {
			int nl = nq;
			memcpy(level, queue, sizeof(int) * (size_t)nl);
			nq = 0;
			stats[2]++;
			for (int a = 0; a < nl; ++a) {
				const int m = level[a];
				outs[m] = decode_cell(bitplane, m, k[m]);
				int far = b[m] < 3 && outs[m].dist < 3 && k_cool(k[m]) == 4 && outs[m].ncool == 4;
				int far_unc = 0;
				if (alt[m] != 0xFFFF) {
					out_t o2 = decode_cell(bitplane, m, alt[m]);
					if (!out_equal(&outs[m], &o2, k[m], alt[m])) BAIL(1, m);
					const int far2 = b[m] < 3 && o2.dist < 3 && k_cool(alt[m]) == 4 && o2.ncool == 4;
					if (far2 != far) { far = 0; far_unc = 1; }
					stats[6]++;
				}
				if (outs[m].dist < p && !(tied == 1 && members == 0 && nl == 1)) BAIL(2, m);
				++members;
				outs[m].w |= far << 8 | far_unc << 9;
			}
			for (int a = 0; a < nl; ++a) { const int m = level[a]; vis[m] = 1; mrnd[m] = round; bpop[m] = b[m]; kused[m] = k[m]; }
			for (int a = 0; a < nl; ++a) {
				const int m = level[a];
				out_t o = outs[m];
				const int far = (o.w >> 8) & 1, far_unc = (o.w >> 9) & 1;
				o.w &= 0xFF;
				outs[m].w = o.w;
				sym_out[m] = (uint8_t)o.sym;
				pos_out[2 * m] = g_pos[2 * m] + k_dx(kused[m]) + o.w % 3 - 1;
				pos_out[2 * m + 1] = g_pos[2 * m + 1] + k_dy(kused[m]) + o.w / 3 - 1;
				++decoded;
				stats[4]++;
				int t[12];
				targets_of(m, t);
				const uint16_t nk = pack_k(o.ndx, o.ndy, (unsigned)o.ncool);
				const int d = o.dist;
				for (int q = 0; q < 12; ++q) {
					const int c = t[q];
					if (c < 0) continue;
					if (q >= 4 && !far) {
						if (far_unc) { unc[nunc].t = c; unc[nunc].k = nk; unc[nunc].d = d; ++nunc; }
						continue;
					}
					if (vis[c]) {
						if (mrnd[c] != round) continue;            /* decoded before this super-round began, in every schedule */
						if (d < bpop[c]) BAIL(3, c);
						if (d == bpop[c] && rnd[c] == round && nk != kused[c]) {
							out_t o2 = decode_cell(bitplane, c, nk);
							if (!out_equal(&outs[c], &o2, kused[c], nk)) BAIL(4, c);
							const int f1 = bpop[c] < 3 && outs[c].dist < 3 && k_cool(kused[c]) == 4 && outs[c].ncool == 4;
							const int f2 = bpop[c] < 3 && o2.dist < 3 && k_cool(nk) == 4 && o2.ncool == 4;
							if (f1 != f2) BAIL(4, c);               /* its horizon offers were already applied as certain */
							stats[6]++;
						}
						continue;
					}
					if (d < b[c]) {
						b[c] = (uint8_t)d; k[c] = nk; rnd[c] = round; alt[c] = 0xFFFF;
						if (d <= p && !queued[c]) { queue[nq++] = c; queued[c] = 1; }
					} else if (d == b[c] && rnd[c] == round && nk != k[c]) {
						if (alt[c] == 0xFFFF) alt[c] = nk;
						else if (alt[c] != nk) BAIL(5, c);
					}
				}
				if (d < p) ended = 1;   /* the single-cell super-round: its offers open a cheaper one */
			}
		}
		if (ended) for (int a = 0; a < nq; ++a) queued[queue[a]] = 0;
		/* horizon offers that some schedules make and others do not: they must not matter */
		for (int u = 0; u < nunc; ++u) {
			const int c = unc[u].t, d = unc[u].d;
			stats[7]++;
			if (vis[c]) {
				if (mrnd[c] != round) continue;
				if (d < bpop[c]) BAIL(6, c);
				if (d == bpop[c] && unc[u].k != kused[c]) BAIL(6, c);
			} else {
				if (d < b[c]) BAIL(6, c);
				if (d == b[c] && unc[u].k != k[c]) BAIL(6, c);
			}
		}
	}
	return 0;
}

/* ---- variant 2: priority-free candidate sets. S(c) = every input (drift, cooldown) some schedule could hand to cell c: the default
 * for a seed, plus whatever any cell that lists c as a target could offer under any of ITS candidate inputs. If every candidate of
 * every cell gives the same symbol / position / new drift / distance (the cooldown it passes on may differ: all variants are
 * propagated), no schedule can produce anything else. Pure over-approximation: sound whatever the heap does.
 * stats: [0] bail code (0 ok, 11 = outcomes differ, 12 = set overflow), [1] sweeps, [2] decodes, [5] cell */
#define SMAX 8
int fm_sets(const uint8_t* bitplane, int prefix, uint8_t* sym_out, int32_t* pos_out, int32_t* stats)
{
	ensure_pos();
	static __thread uint16_t S[NCELLS][SMAX];
	static __thread uint8_t ns[NCELLS], done_n[NCELLS], dirty[NCELLS], vis[NCELLS], b[NCELLS];
	static __thread uint16_t k[NCELLS];
	static __thread out_t first[NCELLS];
	for (int s = 0; s < 8; ++s) stats[s] = 0;
	for (int i = 0; i < NCELLS; ++i) { ns[i] = 0; done_n[i] = 0; dirty[i] = 0; vis[i] = 0; b[i] = 0xFE; k[i] = K_DEFAULT; }
	{
		int decoded = 0;
		heap_t hp = {0, 0, 0};
		uint16_t small_row = TOP_W, last = NCELLS - 1, between = TOP_CELLS;
		hent seeds[8] = {{0, 0}, {(uint16_t)(small_row - 1), 0}, {last, 0}, {(uint16_t)(last - (small_row - 1)), 0},
		                 {between, 1}, {(uint16_t)(between + DIM_X - 1), 1}, {(uint16_t)(last - between), 1},
		                 {(uint16_t)(last - (between + DIM_X - 1)), 1}};
		for (int s = 0; s < 8; ++s) heap_push(&hp, seeds[s]);
		while (decoded < prefix && decoded < NCELLS && hp.n > 0) {
			hent e = heap_pop(&hp);
			int i = e.idx;
			if (vis[i]) continue;
			vis[i] = 1;
			out_t o = decode_cell(bitplane, i, k[i]);
			sym_out[i] = (uint8_t)o.sym;
			pos_out[2 * i] = g_pos[2 * i] + k_dx(k[i]) + o.w % 3 - 1;
			pos_out[2 * i + 1] = g_pos[2 * i + 1] + k_dy(k[i]) + o.w / 3 - 1;
			int t[12];
			targets_of(i, t);
			const int far = b[i] < 3 && o.dist < 3 && k_cool(k[i]) == 4 && o.ncool == 4;
			const uint16_t nk = pack_k(o.ndx, o.ndy, (unsigned)o.ncool);
			for (int q = 0; q < (far ? 12 : 4); ++q) {
				int c = t[q];
				if (c < 0 || vis[c]) continue;
				if (b[c] <= o.dist) continue;
				b[c] = (uint8_t)o.dist; k[c] = nk;
				hent ne = {(uint16_t)c, (uint8_t)o.dist};
				heap_push(&hp, ne);
			}
			++decoded;
		}
		free(hp.v);
		stats[3] = decoded;
	}
	for (int i = 0; i < NCELLS; ++i) if (!vis[i] && (seed_prio(i) != 0xFE || b[i] != 0xFE)) { S[i][0] = k[i]; ns[i] = 1; dirty[i] = 1; }
	int changed = 1;
	while (changed) {
		changed = 0;
		stats[1]++;
		for (int i = 0; i < NCELLS; ++i) {
			if (!dirty[i]) continue;
			dirty[i] = 0;
			if (vis[i]) continue;
			for (int a = done_n[i]; a < ns[i]; ++a) {
				out_t o = decode_cell(bitplane, i, S[i][a]);
				stats[2]++;
				if (a == 0) first[i] = o;
				else {
					const out_t* f = &first[i];
					const int same = f->sym == o.sym && f->dist == o.dist && f->ndx == o.ndx && f->ndy == o.ndy &&
					                 k_dx(S[i][0]) + f->w % 3 == k_dx(S[i][a]) + o.w % 3 && k_dy(S[i][0]) + f->w / 3 == k_dy(S[i][a]) + o.w / 3;
					if (!same) { stats[0] = 11; stats[5] = i; return 11; }
				}
				const uint16_t nk = pack_k(o.ndx, o.ndy, (unsigned)o.ncool);
				int t[12];
				targets_of(i, t);
				const int maybe_far = o.dist < 3 && k_cool(S[i][a]) == 4 && o.ncool == 4;
				for (int q = 0; q < (maybe_far ? 12 : 4); ++q) {
					const int c = t[q];
					if (c < 0 || vis[c]) continue;
					int have = 0;
					for (int z = 0; z < ns[c]; ++z) if (S[c][z] == nk) have = 1;
					if (have) continue;
					if (ns[c] == SMAX) { stats[0] = 12; stats[5] = c; return 12; }
					S[c][ns[c]++] = nk;
					dirty[c] = 1;
					changed = 1;
				}
			}
			done_n[i] = ns[i];
		}
	}
	for (int i = 0; i < NCELLS; ++i) {
		if (!ns[i] || vis[i]) continue;
		sym_out[i] = (uint8_t)first[i].sym;
		pos_out[2 * i] = g_pos[2 * i] + k_dx(S[i][0]) + first[i].w % 3 - 1;
		pos_out[2 * i + 1] = g_pos[2 * i + 1] + k_dy(S[i][0]) + first[i].w / 3 - 1;
	}
	return 0;
}
