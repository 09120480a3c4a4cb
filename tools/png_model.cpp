// png_model.cpp -- CPU model of the DEVICE PNG decoder's algorithm (csrc/png.hip.inc), statement for statement where the arithmetic matters:
// lane-parallel canonical Huffman decode (per-length first / count / base), the lane-parallel input chunks (64 dwords per wavefront, or 16 per
// row of lanes) with their overlap / switch rules and the per-token 64-bit window, the LDS ring (32 / 8 / 4 / 2 KiB) with its flushes and far reads, the periodic overlapped copy (float reciprocal + correction for j mod dist) and the skewed (one row per lane) un-filter.
// Development aid: built by tests/test_png_model.py with g++ and fuzzed against zlib / Pillow on the CPU, so that what is left to find on
// the GPU is plumbing, not arithmetic. Not part of the product and not an oracle for the decode path.
//   g++ -O2 -shared -fPIC -o /tmp/libpng_model.so tools/png_model.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

constexpr int WIN = 32768;
// the ring the device keeps in LDS: 32 KiB (the whole deflate window; flushed 4 KiB at a time) or 8 KiB (flushed 4 KiB at a time; a match from
// further back than RING - 320 reads the bytes the flushes have already put into global memory). Set per call.
int RING = 32768, WMASK = RING - 1, FLUSH = 4096;

enum { PNG_OK = 0, E_BTYPE = -1, E_STORED = -2, E_OVERSUB = -3, E_INCOMPLETE = -4, E_CODE = -5, E_DIST = -6, E_OUTSIZE = -7, E_INPUT = -8, E_HEADER = -9, E_NOEOB = -10,
       E_REPEAT = -11, E_FILTER = -12, E_ADLER = -13, E_MODEL = -99 };

struct Canon {           // what lane L (1..15) holds
	uint32_t first[16];  // first code of length L
	uint32_t count[16];
	int32_t base[16];    // index into sorted[] of the first symbol of length L, minus first[L]
	uint16_t sorted[320];
	int maxlen;
};

// lengths[n] -> canonical tables; zlib's inflate_table rules for what is an error (inftrees.c): over-subscribed always; incomplete unless it is
// a distance (or any non-CODES) set with a single 1-bit code... zlib: "if (left > 0 && (type == CODES || max != 1)) return -1"
int canon_build(Canon& c, const uint8_t* lengths, int n, bool is_codes)
{
	for (int L = 0; L < 16; ++L) { c.count[L] = 0; c.first[L] = 0; c.base[L] = 0; }
	for (int s = 0; s < n; ++s) c.count[lengths[s]]++;
	c.count[0] = 0;
	int maxlen = 0;
	for (int L = 1; L < 16; ++L) if (c.count[L]) maxlen = L;
	c.maxlen = maxlen;
	if (maxlen == 0) return PNG_OK;          // no codes at all: an error only when a code is needed (zlib fills the table with invalid-code entries)
	int left = 1;
	for (int L = 1; L < 16; ++L) { left <<= 1; left -= (int)c.count[L]; if (left < 0) return E_OVERSUB; }
	if (left > 0 && (is_codes || maxlen != 1)) return E_INCOMPLETE;
	uint32_t code = 0, off = 0;
	for (int L = 1; L < 16; ++L) {
		c.first[L] = code;
		c.base[L] = (int32_t)off - (int32_t)code;
		code = (code + c.count[L]) << 1;
		off += c.count[L];
	}
	// sorted by (length, symbol): the device does this with ballots over 64-symbol groups; the order is the same
	uint32_t next[16];
	{ uint32_t o = 0; for (int L = 1; L < 16; ++L) { next[L] = o; o += c.count[L]; } }
	for (int s = 0; s < n; ++s) if (lengths[s]) c.sorted[next[lengths[s]]++] = (uint16_t)s;
	return PNG_OK;
}

// The device reads its input through lane-parallel chunks: 64 dwords per wavefront (k_png_inflate: lane i holds dword cbase + i; the next
// chunk starts 60 further on and is switched to at k >= 62 by the bit buffer, at (bp >> 5) - cbase >= 60 by the symbol loop's 64-bit window)
// or 16 dwords per row of lanes (k_png_inflate4: step 12, window-only). CHUNK = 0 models none of that (plain dword stream), 64 / 16 model
// the chunk bookkeeping as well and fail with E_MODEL whenever an access would leave the lanes of the current chunk.
int CHUNK = 0;
bool g_chunk_fault = false;

struct BitReader {
	const uint32_t* in;     // the stream, dword-aligned start
	size_t nwords, widx;
	uint64_t bb;
	int nb;
	bool overrun;
	// chunk model
	uint32_t cbase = 0, bp = 0;
	bool insym = false;          // inside a block's symbol loop: one chunk check + one 64-bit window per token, fields come out of that window
	uint64_t Rtok = 0;
	uint32_t tokbase = 0;
	void begin_symbols() { if (CHUNK) { if (CHUNK == 64) bp = bitpos(); insym = true; } }
	void end_symbols() { if (insym) { insym = false; if (CHUNK == 64) seat_bit(bp); } }
	void token() { if (insym) { ensure_window(bp); Rtok = window(bp); tokbase = bp; } else refill(); }
	uint32_t at(size_t i) const { return i < nwords ? in[i] : 0u; }
	uint32_t lane_of(size_t i, int lanes)          // dword i out of the current chunk
	{
		if (i < cbase || i - cbase >= (size_t)lanes) { g_chunk_fault = true; return 0; }
		return at(i);
	}
	uint32_t next_dword()
	{
		if (widx > nwords + 2) overrun = true;
		if (CHUNK == 64) {
			if (widx - cbase >= 62) cbase += 60;                 // ensure(): cur = nxt
			const uint32_t v = lane_of(widx, 64);
			++widx;
			return v;
		}
		return at(widx++);
	}
	uint64_t window(uint32_t p)                     // the 64 bits at bit position p (three lanes of the current chunk)
	{
		const size_t kk = p >> 5;
		const uint32_t sh = p & 31u, lanes = CHUNK == 16 ? 16 : 64;
		const uint32_t d0 = lane_of(kk, lanes), d1 = lane_of(kk + 1, lanes), d2 = lane_of(kk + 2, lanes);
		const uint32_t lo = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh), hi = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
		return ((uint64_t)hi << 32) | lo;
	}
	void ensure_window(uint32_t p)                  // the symbol loop's (and Bits4's) chunk check
	{
		const uint32_t step = CHUNK == 16 ? 12 : 60;
		if ((p >> 5) - cbase >= step) cbase += step;
	}
	void refill() { if (!insym && CHUNK != 16 && nb <= 32) { bb |= (uint64_t)next_dword() << nb; nb += 32; } }      // afterwards nb >= 33: a caller may take up to 32 bits
	uint32_t peek(int n)
	{
		if (insym) { const uint32_t sh = bp - tokbase; if (sh + (uint32_t)n > 64u) g_chunk_fault = true; return (uint32_t)(Rtok >> (sh & 63u)) & ((1u << n) - 1u); }
		if (CHUNK == 16) { ensure_window(bp); return (uint32_t)window(bp) & ((1u << n) - 1u); }
		return (uint32_t)(bb & ((1ull << n) - 1ull));
	}
	void drop(int n) { if (insym || CHUNK == 16) { bp += (uint32_t)n; if ((bp >> 5) > nwords + 2) overrun = true; } else { bb >>= n; nb -= n; } }
	uint32_t get(int n) { const uint32_t v = peek(n); drop(n); return v; }
	void align() { if (CHUNK == 16) bp = (bp + 7u) & ~7u; else drop(nb & 7); }
	size_t bytepos() const { return CHUNK == 16 ? (size_t)(bp >> 3) : widx * 4 - (size_t)(nb / 8); }
	void seat_byte(size_t np)                       // continue at byte np of the stream (after a stored block)
	{
		if (CHUNK == 16) { bp = (uint32_t)(np * 8); cbase = bp >> 5; return; }
		widx = np / 4; cbase = (uint32_t)widx; bb = 0; nb = 0;
		refill();
		drop((int)(8 * (np % 4)));
	}
	uint32_t bitpos() const { return CHUNK == 16 ? bp : (uint32_t)(widx * 32 - (size_t)nb); }
	void seat_bit(uint32_t p)                       // the symbol loop hands the position back to the bit buffer
	{
		if (CHUNK == 16) { bp = p; return; }
		widx = p >> 5; bb = 0; nb = 0;
		refill();
		drop((int)(p & 31u));
	}
};

uint32_t rev15(uint32_t v)   // bit-reverse the low 15 bits (the device: v_bfrev_b32 >> 17)
{
	uint32_t r = 0;
	for (int i = 0; i < 15; ++i) r |= ((v >> i) & 1u) << (14 - i);
	return r;
}

// one symbol: every lane L tests "the first L bits are a code of length L"; the lowest such L wins
int canon_decode(const Canon& c, BitReader& br, int* sym)
{
	const uint32_t r = rev15(br.peek(15));
	for (int L = 1; L <= 15; ++L) {                 // (lanes)
		const uint32_t codeL = r >> (15 - L);
		if (codeL - c.first[L] < c.count[L]) {
			*sym = c.sorted[(int32_t)codeL + c.base[L]];
			br.drop(L);
			return PNG_OK;
		}
	}
	return E_CODE;
}

const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

unsigned long long g_stats[8];   // literals, matches, match bytes, blocks, stored bytes, overlapped matches, dynamic blocks, all-offsets turns

struct Out {
	uint8_t win[WIN];
	uint8_t* glob;
	uint32_t op, cap;
	void flush_to(uint32_t upto)      // [flushed, upto) ring -> global
	{
		for (uint32_t q = flushed; q < upto; q += FLUSH) adler_segment(q, upto - q < (uint32_t)FLUSH ? upto : q + FLUSH);
		for (uint32_t p = flushed; p < upto; ++p) glob[p] = win[p & WMASK];
		flushed = upto;
	}
	uint32_t flushed;
	uint32_t a1 = 1, a2 = 0;            // Adler-32 of everything flushed so far
	void adler_segment(uint32_t from, uint32_t upto)
	{
		// per segment of n <= 4096 bytes: A' = A + sum d_i, B' = B + n*A + sum (n - i) d_i  (both sums < 2^32: lanes add their share, one wave reduction)
		const uint32_t n = upto - from;
		uint32_t s1 = 0, s2 = 0;
		for (uint32_t i = 0; i < n; ++i) { const uint32_t d = win[(from + i) & WMASK]; s1 += d; s2 += (n - i) * d; }
		a2 = (uint32_t)(((uint64_t)a2 + (uint64_t)n * a1 + s2) % 65521u);
		a1 = (a1 + s1) % 65521u;
	}
	void after(uint32_t before) { if ((op ^ before) & ~(uint32_t)(FLUSH - 1)) flush_to(op & ~(uint32_t)(FLUSH - 1)); }
};

// j mod d for 0 <= j < 322, 1 <= d < 32768, the way the device does it (v_rcp_f32 is not exact: model it with a perturbed reciprocal too)
inline uint32_t modsmall(uint32_t j, uint32_t d, float rcp)
{
	int q = (int)((float)j * rcp);
	int r = (int)j - q * (int)d;
	if (r < 0) r += (int)d;
	if (r >= (int)d) r -= (int)d;
	return (uint32_t)r;
}

// ---- the all-offsets turn of k_png_inflate<RING, true> (csrc/png.hip.inc, par_build and the MODE == 2 turn): every lane o decodes the token that
// would start at bit o of the window (9-bit tables: a code of more than 9 bits is "not decodable here"), the chain from offset 0 picks the real
// ones, the turn's literals are stored at once, short runs whose byte is final are filled by their own lanes, the other matches copied in
// order. Besides the bytes this checks what the kernel relies on: that no source byte of a match has been overwritten by something the turn
// stored AHEAD of it (the ring holds RING bytes), and that a far match's source has been flushed. Returns 0 (turn done), -1 (the token at bp
// is for the one-token path) or an error code.
int PARMODE = 0;

bool canon9(const Canon& c, uint32_t bits9, int* sym, int* len)
{
	uint32_t r = 0;
	for (int i = 0; i < 9; ++i) r |= ((bits9 >> i) & 1u) << (8 - i);
	for (int L = 9; L >= 1; --L) {
		const uint32_t codeL = r >> (9 - L);
		if (codeL - c.first[L] < c.count[L]) { *sym = c.sorted[(int32_t)codeL + c.base[L]]; *len = L; return true; }
	}
	return false;
}

int par_turn(BitReader& br, const Canon& ll, const Canon& dd, Out& o, float rcp_scale)
{
	br.ensure_window(br.bp);
	const uint32_t kk = br.bp >> 5, sh0 = br.bp & 31u;
	bool lit[64], mat[64];
	uint32_t val[64], mlen[64], mdist[64], link[64], outlen[64];
	for (int l = 0; l < 64; ++l) {
		const uint32_t bit0 = sh0 + (uint32_t)l, w0 = kk + (bit0 >> 5), sh = bit0 & 31u;
		const uint32_t d0 = br.lane_of(w0, 64), d1 = br.lane_of(w0 + 1, 64), d2 = br.lane_of(w0 + 2, 64);
		const uint64_t B = (uint64_t)(uint32_t)((((uint64_t)d1 << 32) | d0) >> sh) | ((uint64_t)(uint32_t)((((uint64_t)d2 << 32) | d1) >> sh) << 32);
		lit[l] = mat[l] = false; val[l] = mlen[l] = mdist[l] = 0; outlen[l] = 0;
		uint32_t bits = 0;
		int sym, L1;
		if (canon9(ll, (uint32_t)B & 511u, &sym, &L1)) {
			if (sym < 256) { lit[l] = true; val[l] = (uint32_t)sym; bits = (uint32_t)L1; outlen[l] = 1; }
			else if (sym >= 257 && sym <= 285) {
				const uint32_t x1 = LEXT[sym - 257];
				const uint32_t len = LBASE[sym - 257] + ((uint32_t)(B >> L1) & ((1u << x1) - 1u));
				const uint64_t B2 = B >> ((uint32_t)L1 + x1);
				int ds, L2;
				if (canon9(dd, (uint32_t)B2 & 511u, &ds, &L2) && ds <= 29) {
					const uint32_t x2 = DEXT[ds];
					mat[l] = true; mlen[l] = len; outlen[l] = len;
					mdist[l] = DBASE[ds] + ((uint32_t)(B2 >> L2) & ((1u << x2) - 1u));
					bits = (uint32_t)L1 + x1 + (uint32_t)L2 + x2;
				}
			}
		}
		link[l] = (lit[l] || mat[l]) ? (uint32_t)l + bits : 128u;
	}
	if (g_chunk_fault) return E_MODEL;
	bool vis[64] = {false};
	uint32_t off = 0, t = 0;
	for (;;) {
		t = link[off];
		if (t >= 64u) break;
		vis[off] = true;
		off = t;
	}
	if (!(t & 128u)) { vis[off] = true; off = t; }
	bool any = false;
	for (int l = 0; l < 64; ++l) any = any || vis[l];
	if (!any) return -1;
	uint32_t incl[64], acc = 0;
	for (int l = 0; l < 64; ++l) { acc += vis[l] ? outlen[l] : 0u; incl[l] = acc; }
	for (int l = 0; l < 64; ++l)
		if (vis[l] && incl[l] > 1024u) {
			if (l != 0) { for (int k = l; k < 64; ++k) vis[k] = false; off = (uint32_t)l; }
			else return E_MODEL;
			break;
		}
	int lastv = 0;
	for (int l = 0; l < 64; ++l) if (vis[l]) lastv = l;
	const uint32_t before = o.op, total = incl[lastv];
	if (before + total > o.cap) return E_OUTSIZE;
	for (int l = 0; l < 64; ++l) if (vis[l] && mat[l] && mdist[l] > before + (incl[l] - outlen[l])) return E_DIST;
	if (RING >= 32768)
		for (int l = 0; l < 64; ++l) if (vis[l] && mat[l] && mdist[l] > (uint32_t)RING - 2048u) return -1;     // token by token
	const uint32_t nearlim = (uint32_t)RING - 2048u, turn_end = before + total;
	for (int l = 0; l < 64; ++l) if (vis[l] && lit[l]) { o.win[(before + incl[l] - 1u) & WMASK] = (uint8_t)val[l]; g_stats[0]++; }
	// short runs whose byte is final: all lanes read, then all write
	bool own[64];
	uint8_t fill[64];
	int prev = -1;
	for (int l = 0; l < 64; ++l) {
		own[l] = vis[l] && mat[l] && mdist[l] == 1u && mlen[l] <= 32u && (prev < 0 || lit[prev]);
		if (own[l]) fill[l] = o.win[(before + incl[l] - outlen[l] - 1u) & WMASK];
		if (vis[l]) prev = l;
	}
	for (int l = 0; l < 64; ++l)
		if (own[l]) { for (uint32_t j = 0; j < mlen[l]; ++j) o.win[(before + incl[l] - outlen[l] + j) & WMASK] = fill[l]; g_stats[1]++; g_stats[2] += mlen[l]; }
	for (int l = 0; l < 64; ++l) {
		if (!(vis[l] && mat[l]) || own[l]) continue;
		const uint32_t at = before + incl[l] - outlen[l], len = mlen[l], dist = mdist[l], src0 = at - dist;
		g_stats[1]++; g_stats[2] += len; if (dist < len) g_stats[5]++;
		if (dist == 1u) {
			const uint8_t b = o.win[src0 & WMASK];
			for (uint32_t j = 0; j < len; ++j) o.win[(at + j) & WMASK] = b;
		} else if (dist >= len && RING < 32768 && dist > nearlim) {
			if (src0 + len > o.flushed) return E_MODEL;                          // a far source that has not been flushed
			for (uint32_t j = 0; j < len; ++j) o.win[(at + j) & WMASK] = o.glob[src0 + j];
		} else {
			// in the ring: nothing this turn stored ahead (up to turn_end) may have landed on a source byte
			if (turn_end - src0 > (uint32_t)RING) return E_MODEL;
			const float rcp = rcp_scale / (float)dist;
			for (uint32_t c0 = 0; c0 < len; c0 += 64) {
				uint8_t tmp[64];
				const uint32_t m = len - c0 < 64 ? len - c0 : 64;
				for (uint32_t q = 0; q < m; ++q) { const uint32_t j = c0 + q, r = dist >= len ? j : modsmall(j, dist, rcp); tmp[q] = o.win[(src0 + r) & WMASK]; }
				for (uint32_t q = 0; q < m; ++q) o.win[(at + c0 + q) & WMASK] = tmp[q];
			}
		}
	}
	br.bp += off;
	if ((br.bp >> 5) > br.nwords + 2) br.overrun = true;
	o.op = before + total;
	o.after(before);
	g_stats[7]++;                 // turns taken by this path
	return 0;
}

int inflate_model(const uint8_t* zs, size_t zlen, uint8_t* out, uint32_t expect, float rcp_scale)
{
	if (zlen < 2) return E_HEADER;
	if ((zs[0] & 15) != 8 || (zs[0] >> 4) > 7 || ((zs[0] << 8) | zs[1]) % 31 != 0 || (zs[1] & 0x20)) return E_HEADER;
	std::vector<uint32_t> words((zlen + 3) / 4 + 1, 0);
	std::memcpy(words.data(), zs, zlen);
	BitReader br{words.data(), (zlen + 3) / 4, 0, 0, 0, false};
	g_chunk_fault = false;
	br.refill();
	br.drop(16);                                  // CMF, FLG
	static Out o;
	o.glob = out; o.op = 0; o.cap = expect; o.flushed = 0; o.a1 = 1; o.a2 = 0;
	for (auto& v : g_stats) v = 0;
	Canon cl, ll, dd;
	uint8_t lengths[320];
	for (;;) {
		br.refill();
		const uint32_t bfinal = br.get(1), btype = br.get(2);
		if (btype == 3) return E_BTYPE;
		g_stats[3]++; if (btype == 2) g_stats[6]++;
		if (btype == 0) {
			br.align();
			br.refill();
			const uint32_t len = br.get(16);
			br.refill();
			const uint32_t nlen = br.get(16);
			if ((len ^ nlen) != 0xFFFFu) return E_STORED;
			// the bytes follow at byte position 4 * widx - nb / 8 of the stream
			const size_t bpos = br.bytepos();
			if (bpos + len > zlen) return E_INPUT;
			if (o.op + len > o.cap) return E_OUTSIZE;
			g_stats[4] += len;
			for (uint32_t c0 = 0; c0 < len; c0 += 64) {          // (64 lanes per pass)
				const uint32_t before = o.op, m = len - c0 < 64 ? len - c0 : 64;
				for (uint32_t l = 0; l < m; ++l) o.win[(o.op + l) & WMASK] = zs[bpos + c0 + l];
				o.op += m;
				o.after(before);
			}
			br.seat_byte(bpos + len);
		} else {
			if (btype == 1) {
				for (int s = 0; s < 144; ++s) lengths[s] = 8;
				for (int s = 144; s < 256; ++s) lengths[s] = 9;
				for (int s = 256; s < 280; ++s) lengths[s] = 7;
				for (int s = 280; s < 288; ++s) lengths[s] = 8;
				int rc = canon_build(ll, lengths, 288, false);
				if (rc) return rc;
				for (int s = 0; s < 32; ++s) lengths[s] = 5;              // (zlib's fixed distance table has 32 entries too; 30 and 31 are invalid when met)
				rc = canon_build(dd, lengths, 32, false);
				if (rc) return rc;
			} else {
				br.refill();
				const int hlit = (int)br.get(5) + 257, hdist = (int)br.get(5) + 1, hclen = (int)br.get(4) + 4;
				if (hlit > 286 || hdist > 30) return E_HEADER;
				uint8_t cll[19];
				std::memset(cll, 0, sizeof cll);
				for (int k = 0; k < hclen; ++k) { br.refill(); cll[CLORDER[k]] = (uint8_t)br.get(3); }
				int rc = canon_build(cl, cll, 19, true);
				if (rc) return rc;
				int n = 0;
				while (n < hlit + hdist) {
					br.refill();
					int sym;
					rc = canon_decode(cl, br, &sym);
					if (rc) return rc;
					if (sym < 16) lengths[n++] = (uint8_t)sym;
					else {
						int rep, val = 0;
						if (sym == 16) { if (n == 0) return E_REPEAT; val = lengths[n - 1]; rep = 3 + (int)br.get(2); }
						else if (sym == 17) rep = 3 + (int)br.get(3);
						else rep = 11 + (int)br.get(7);
						if (n + rep > hlit + hdist) return E_REPEAT;
						while (rep--) lengths[n++] = (uint8_t)val;
					}
				}
				if (lengths[256] == 0) return E_NOEOB;
				rc = canon_build(ll, lengths, hlit, false);
				if (rc) return rc;
				rc = canon_build(dd, lengths + hlit, hdist, false);
				if (rc) return rc;
			}
			br.begin_symbols();
			for (;;) {
				if (PARMODE && CHUNK == 64) {
					const int pr = par_turn(br, ll, dd, o, rcp_scale);
					if (pr == 0) continue;
					if (pr > 0) return pr;
				}
				br.token();
				int sym;
				int rc = canon_decode(ll, br, &sym);
				if (rc) return rc;
				if (sym < 256) {
					if (o.op >= o.cap) return E_OUTSIZE;
					const uint32_t before = o.op;
					const uint32_t l1 = br.insym ? br.bp - br.tokbase : 99u;
					const int FB = CHUNK == 16 ? 7 : 8;           // k_png_inflate4's table is indexed by seven bits, k_png_inflate's by eight
					if (l1 > (uint32_t)FB) {
						o.win[o.op & WMASK] = (uint8_t)sym;
						g_stats[0]++;
						o.op += 1;
						o.after(before);
						continue;
					}
					// a literal burst (the kernels' fast_build + walk): every place of the turn's window where a short literal could start is
					// looked up; the chain 0 -> len(0) -> ... is followed while it meets literals of <= 8 bits, up to the end of the image
					const uint32_t limit = CHUNK == 16 ? 48u : 56u, room = o.cap - before;
					uint32_t off = 0, cnt = 0;
					while (off < limit && cnt < room) {
						const uint32_t b = (uint32_t)(br.Rtok >> off) & ((1u << FB) - 1u);
						uint32_t r8 = 0;
						for (int i = 0; i < FB; ++i) r8 |= ((b >> i) & 1u) << (FB - 1 - i);
						uint32_t f = 0;
						for (int L = FB; L >= 1; --L) {
							const uint32_t codeL = r8 >> (FB - L);
							if (codeL - ll.first[L] < ll.count[L]) {
								const uint32_t e = ll.sorted[(int32_t)codeL + ll.base[L]];
								f = e < 256u ? ((uint32_t)L << 9) | e : 0u;
							}
						}
						if (!f) break;
						o.win[(before + cnt) & WMASK] = (uint8_t)f;
						off += f >> 9;
						++cnt;
						g_stats[0]++;
					}
					if (cnt == 0) return E_MODEL;                  // (the first literal is a short one: the table must have it)
					br.bp = br.tokbase + off;
					if ((br.bp >> 5) > br.nwords + 2) br.overrun = true;
					o.op = before + cnt;
					o.after(before);
					if (off > 16u) continue;
					// the token behind the burst out of the same window, if it is a match
					const uint32_t keep = br.bp;
					if (canon_decode(ll, br, &sym) != PNG_OK) continue;       // (nothing consumed: the next turn reports it)
					if (sym <= 256) { br.bp = keep; continue; }
				}
				if (sym == 256) { br.end_symbols(); break; }
				if (sym > 285) return E_CODE;
				const uint32_t len = LBASE[sym - 257] + br.get(LEXT[sym - 257]);   // <= 15 + 5 bits since the refill
				br.refill();
				int ds;
				rc = canon_decode(dd, br, &ds);
				if (rc) return rc;
				if (ds > 29) return E_CODE;
				const uint32_t dist = DBASE[ds] + br.get(DEXT[ds]);                   // <= 15 + 13
				if (dist > o.op) return E_DIST;
				if (o.op + len > o.cap) return E_OUTSIZE;
				const float rcp = rcp_scale / (float)dist;
				g_stats[1]++; g_stats[2] += len; if (dist < len) g_stats[5]++;
				const uint32_t before = o.op;
				if (dist > (uint32_t)RING - 320u) {
					// far: every source byte has been flushed (op - flushed < FLUSH + 258, sources end below op - RING + 578) and dist >= len
					if (before - dist + len > o.flushed || dist < len) return E_MODEL;
					for (uint32_t j = 0; j < len; ++j) o.win[(before + j) & WMASK] = o.glob[before - dist + j];
				} else
				for (uint32_t c0 = 0; c0 < len; c0 += 64) {
					uint8_t tmp[64];
					const uint32_t m = len - c0 < 64 ? len - c0 : 64;
					for (uint32_t l = 0; l < m; ++l) {             // all lanes read ...
						const uint32_t j = c0 + l, r = dist >= len ? j : modsmall(j, dist, rcp);
						tmp[l] = o.win[(before - dist + r) & WMASK];
					}
					for (uint32_t l = 0; l < m; ++l) o.win[(before + c0 + l) & WMASK] = tmp[l];   // ... then all lanes write
				}
				o.op += len;
				o.after(before);
			}
		}
		if (bfinal) break;
	}
	if (br.overrun) return E_INPUT;
	if (g_chunk_fault) return E_MODEL;
	o.flush_to(o.op);
	if (o.op != expect) return E_OUTSIZE;
	// the Adler-32 of the inflated bytes follows the last block at the next byte boundary, big-endian (RFC 1950)
	br.align();
	uint32_t want = 0;
	for (int k = 0; k < 4; ++k) { br.refill(); want = (want << 8) | br.get(8); }
	if (br.overrun) return E_INPUT;
	if (g_chunk_fault) return E_MODEL;
	return want == ((o.a2 << 16) | o.a1) ? PNG_OK : E_ADLER;
}

inline int paeth(int a, int b, int c)
{
	const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// the skewed un-filter: lane l owns row 64*band + l and is l pixels behind the lane above; "up" travels lane to lane, the band's last row
// goes through a boundary row in memory. raw = h * (1 + w*bpp) filtered bytes, un-filtered in `unf` (h * w * bpp)
int unfilter_model(const uint8_t* raw, uint8_t* unf, int w, int h, int bpp)
{
	const int rb = w * bpp;
	std::vector<uint8_t> boundary((size_t)rb, 0);
	for (int band = 0; band * 64 < h; ++band) {
		uint32_t prev_out[64], prev_b[64], a[64];
		for (int l = 0; l < 64; ++l) prev_out[l] = prev_b[l] = a[l] = 0;
		std::vector<uint8_t> newboundary((size_t)rb, 0);
		for (int t = 0; t < w + 63; ++t) {
			uint32_t cur_out[64], cur_b[64];
			for (int l = 0; l < 64; ++l) {
				const int row = band * 64 + l, x = t - l;
				cur_out[l] = 0; cur_b[l] = 0;
				uint32_t b;                                   // packed channels of the pixel above
				if (l > 0) b = prev_out[l - 1];
				else {
					b = 0;
					if (band > 0 && x >= 0 && x < w) for (int ch = 0; ch < bpp; ++ch) b |= (uint32_t)boundary[(size_t)x * bpp + ch] << (8 * ch);
				}
				cur_b[l] = b;
				if (row >= h || x < 0 || x >= w) continue;
				const uint32_t c = prev_b[l];                 // the pixel above-left = what was "above" one step ago
				const uint8_t ft = raw[(size_t)row * (rb + 1)];
				if (ft > 4) return E_FILTER;
				uint32_t o = 0;
				for (int ch = 0; ch < bpp; ++ch) {
					const int f = raw[(size_t)row * (rb + 1) + 1 + (size_t)x * bpp + ch];
					const int av = x > 0 ? (int)((a[l] >> (8 * ch)) & 255u) : 0, bv = (int)((b >> (8 * ch)) & 255u), cv = x > 0 ? (int)((c >> (8 * ch)) & 255u) : 0;
					int pred = 0;
					switch (ft) { case 1: pred = av; break; case 2: pred = bv; break; case 3: pred = (av + bv) >> 1; break; case 4: pred = paeth(av, bv, cv); break; default: break; }
					o |= (uint32_t)((f + pred) & 255) << (8 * ch);
				}
				cur_out[l] = o;
				a[l] = o;
				for (int ch = 0; ch < bpp; ++ch) unf[((size_t)row * w + x) * bpp + ch] = (uint8_t)(o >> (8 * ch));
				if (l == 63) for (int ch = 0; ch < bpp; ++ch) newboundary[(size_t)x * bpp + ch] = (uint8_t)(o >> (8 * ch));
			}
			for (int l = 0; l < 64; ++l) { prev_out[l] = cur_out[l]; prev_b[l] = cur_b[l]; }
		}
		boundary.swap(newboundary);
	}
	return PNG_OK;
}


// the device's fast path (unfilter_image4, widths that are a multiple of 4): the same skew in units of four pixels, with the device's byte
// arithmetic -- Sub / Up / Average on all channels of a dword at once, Paeth on two 16-bit lanes per register -- restated operation for operation
inline uint32_t add_bytes(uint32_t x, uint32_t y) { return ((x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu)) ^ ((x ^ y) & 0x80808080u); }
inline uint32_t avg_bytes(uint32_t x, uint32_t y) { return (x & y) + (((x ^ y) & 0xfefefefeu) >> 1); }
inline uint32_t paeth_packed(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t out = 0;
	for (int h = 0; h < 2; ++h) {
		// bytes (2h, 2h + 1) as two 16-bit lanes
		int16_t A[2], B[2], C[2], P[2];
		for (int k = 0; k < 2; ++k) { A[k] = (int16_t)((a >> (8 * (2 * h + k))) & 255u); B[k] = (int16_t)((b >> (8 * (2 * h + k))) & 255u); C[k] = (int16_t)((c >> (8 * (2 * h + k))) & 255u); }
		for (int k = 0; k < 2; ++k) {
			const int16_t d1 = (int16_t)(B[k] - C[k]), d2 = (int16_t)(A[k] - C[k]), d3 = (int16_t)(d1 + d2);
			const int16_t pa = d1 > (int16_t)-d1 ? d1 : (int16_t)-d1, pb = d2 > (int16_t)-d2 ? d2 : (int16_t)-d2, pc = d3 > (int16_t)-d3 ? d3 : (int16_t)-d3;
			const uint16_t n1 = (uint16_t)((int16_t)(pb - pa) >> 15), n2 = (uint16_t)((int16_t)(pc - pa) >> 15), n3 = (uint16_t)((int16_t)(pc - pb) >> 15);
			const uint16_t m = n1 | n2, bc = (uint16_t)(((uint16_t)B[k] & ~n3) | ((uint16_t)C[k] & n3));
			P[k] = (int16_t)(((uint16_t)A[k] & ~m) | (bc & m));
		}
		out |= ((uint32_t)(P[0] & 255) << (8 * (2 * h))) | ((uint32_t)(P[1] & 255) << (8 * (2 * h + 1)));
	}
	return out;
}

int unfilter_model4(const uint8_t* raw, uint8_t* unf, int w, int h, int bpp)
{
	const int rb = w * bpp, nblk = w / 4;
	const uint32_t pmask = bpp == 4 ? 0xFFFFFFFFu : (1u << (8 * bpp)) - 1u;
	std::vector<uint32_t> boundary((size_t)w, 0);
	for (int band = 0; band * 64 < h; ++band) {
		uint32_t po[64][4], a[64], upprev3[64];
		for (int l = 0; l < 64; ++l) { a[l] = upprev3[l] = 0; for (int k = 0; k < 4; ++k) po[l][k] = 0; }
		bool any_paeth = false;
		for (int l = 0; l < 64; ++l) if (band * 64 + l < h && raw[(size_t)(band * 64 + l) * (rb + 1)] == 4) any_paeth = true;
		for (int T = 0; T < nblk + 63; ++T) {
			uint32_t npo[64][4], nup3[64];
			std::vector<std::pair<int, uint32_t>> bwrites;
			for (int l = 0; l < 64; ++l) {
				const int row = band * 64 + l, q = T - l;
				const bool rowok = row < h, active = rowok && q >= 0 && q < nblk;
				uint32_t up[4];
				for (int k = 0; k < 4; ++k) up[k] = l > 0 ? po[l - 1][k] : 0u;          // wave_shr:1 of last step's outputs
				if (l == 0 && band > 0 && q < nblk) for (int k = 0; k < 4; ++k) up[k] = boundary[(size_t)4 * q + k];
				const uint32_t ft = rowok ? raw[(size_t)row * (rb + 1)] : 0u;
				if (ft > 4) return E_FILTER;
				for (int k = 0; k < 4; ++k) {
					uint32_t f = 0;
					if (active) for (int ch = 0; ch < bpp; ++ch) f |= (uint32_t)raw[(size_t)row * (rb + 1) + 1 + ((size_t)4 * q + k) * bpp + ch] << (8 * ch);
					const uint32_t b = up[k], c = k == 0 ? upprev3[l] : up[k - 1];
					uint32_t pred = 0;
					pred = ft == 1u ? a[l] : pred;
					pred = ft == 2u ? b : pred;
					pred = ft == 3u ? avg_bytes(a[l], b) : pred;
					if (any_paeth && ft == 4u) {
						if (bpp < 3) { pred = 0; for (int ch = 0; ch < bpp; ++ch) pred |= (uint32_t)paeth((int)((a[l] >> (8 * ch)) & 255u), (int)((b >> (8 * ch)) & 255u), (int)((c >> (8 * ch)) & 255u)) << (8 * ch); }
						else pred = paeth_packed(a[l], b, c);
					}
					uint32_t o = add_bytes(f, pred) & pmask;
					o = active ? o : 0u;
					a[l] = o;
					npo[l][k] = o;
					if (active) for (int ch = 0; ch < bpp; ++ch) unf[((size_t)row * w + 4 * q + k) * bpp + ch] = (uint8_t)(o >> (8 * ch));
					if (active && l == 63) bwrites.emplace_back(4 * q + k, o);
				}
				nup3[l] = up[3];
			}
			for (int l = 0; l < 64; ++l) { upprev3[l] = nup3[l]; for (int k = 0; k < 4; ++k) po[l][k] = npo[l][k]; }
			for (auto& bw : bwrites) boundary[(size_t)bw.first] = bw.second;
		}
	}
	return PNG_OK;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

}  // namespace

// PNG (8-bit gray / RGB / RGBA / palette, non-interlaced) -> un-filtered samples (h*w*bpp); returns 0 or a negative code; *pw,*ph,*pbpp set
extern "C" void png_model_chunk(int chunk) { CHUNK = chunk; }
extern "C" void png_model_par(int on) { PARMODE = on; }

extern "C" void png_model_ring(int ring) { RING = ring; WMASK = ring - 1; FLUSH = ring >= 32768 ? 4096 : ring / 2; }

extern "C" int png_model_decode(const uint8_t* png, size_t len, uint8_t* unf, size_t cap, unsigned* pw, unsigned* ph, unsigned* pbpp, float rcp_scale)
{
	if (len < 33 || std::memcmp(png, "\x89PNG\r\n\x1a\n", 8) != 0) return E_HEADER;
	size_t pos = 8;
	unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat;
	while (pos + 12 <= len) {
		const uint32_t clen = be32(png + pos);
		if ((size_t)clen > len - pos - 12) return E_HEADER;
		if (!std::memcmp(png + pos + 4, "IHDR", 4)) { w = be32(png + pos + 8); h = be32(png + pos + 12); depth = png[pos + 16]; ctype = png[pos + 17]; interlace = png[pos + 20]; }
		else if (!std::memcmp(png + pos + 4, "IDAT", 4)) idat.insert(idat.end(), png + pos + 8, png + pos + 8 + clen);
		else if (!std::memcmp(png + pos + 4, "IEND", 4)) break;
		pos += 12 + (size_t)clen;
	}
	if (!w || !h || depth != 8 || interlace) return E_HEADER;
	const int bpp = ctype == 2 ? 3 : ctype == 6 ? 4 : (ctype == 0 || ctype == 3) ? 1 : 0;
	if (!bpp) return E_HEADER;
	*pw = w; *ph = h; *pbpp = (unsigned)bpp;
	if (!unf) return 0;
	if (cap < (size_t)w * h * bpp) return E_OUTSIZE;
	std::vector<uint8_t> raw((size_t)h * ((size_t)w * bpp + 1));
	int rc = inflate_model(idat.data(), idat.size(), raw.data(), (uint32_t)raw.size(), rcp_scale);
	if (rc) return rc;
	return (w % 4 == 0) ? unfilter_model4(raw.data(), unf, (int)w, (int)h, bpp) : unfilter_model(raw.data(), unf, (int)w, (int)h, bpp);
}

extern "C" void png_model_stats(unsigned long long* out) { for (int k = 0; k < 8; ++k) out[k] = g_stats[k]; }

extern "C" int png_model_inflate(const uint8_t* zs, size_t zlen, uint8_t* out, uint32_t expect, float rcp_scale) { return inflate_model(zs, zlen, out, expect, rcp_scale); }
