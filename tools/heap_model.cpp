// heap_model.cpp -- CPU model of the lane-parallel heap operations of the exact flood replay (csrc/k2b_flood.hip.inc: WaveHeap::push,
// WaveHeap::pop with five levels per round, WaveHeap7::pop7 with six, WaveHeap7::push4) and of k_flood3's rule for the entry that is on top after a step's
// pushes, run against libstdc++'s own std::push_heap / std::pop_heap -- the thing FloodDecodePositions' std::priority_queue is made of
// (FloodDecodePositions.h:18-28,48). Sixty-four "lanes" are arrays here; ballots are bit masks. Development aid (what is left to find on the
// GPU is plumbing, not arithmetic); not part of the product and not an oracle for the decode path. Built and driven by tests/test_heap_model.py.
//   g++ -O2 -shared -fPIC -o libheap_model.so tools/heap_model.cpp
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

namespace {

struct Cmp { bool operator()(uint32_t a, uint32_t b) const { return (a >> 25) > (b >> 25); } };   // PrioCompare: a.prio > b.prio

inline int clz32(uint32_t v) { return __builtin_clz(v); }
inline int clz64(uint64_t v) { return __builtin_clzll(v); }

struct WaveHeapModel {
	std::vector<uint32_t> H;   // the LDS array (plus spill: one flat array here)
	int n = 0;
	uint64_t ancm[64], dirm[64];
	WaveHeapModel() : H(1 << 18, 0u)
	{
		for (int lane = 0; lane < 64; ++lane) {
			ancm[lane] = dirm[lane] = 0;
			if (lane >= 1 && lane < 63) {
				const unsigned node = (unsigned)lane + 1u;
				for (unsigned c = node; c > 1u; c >>= 1) {
					const unsigned p = c >> 1;
					ancm[lane] |= 1ull << (p - 1u);
					if (c & 1u) dirm[lane] |= 1ull << (p - 1u);
				}
			}
		}
	}
	uint32_t get(int i, bool on = true) const { return H[on ? i : 0]; }

	void push(uint32_t e)
	{
		const int pos = n++;
		const int depth = 31 - clz32((unsigned)pos + 1u);
		int a[64]; uint32_t v[64]; bool anc[64];
		uint64_t stop = 0;
		for (int lane = 0; lane < 64; ++lane) {
			a[lane] = (int)(((unsigned)pos + 1u) >> (lane < 31 ? lane : 31)) - 1;
			anc[lane] = lane >= 1 && lane <= depth;
			v[lane] = get(a[lane], anc[lane]);
			if (anc[lane] && !((v[lane] >> 25) > (e >> 25))) stop |= 1ull << lane;
		}
		const int m = stop ? __builtin_ctzll(stop) - 1 : depth;
		std::vector<std::pair<int, uint32_t>> wr;
		for (int lane = 0; lane < 64; ++lane) {
			const uint32_t vnext = lane < 63 ? v[lane + 1] : 0u;   // from_right_lane
			if (lane <= m) wr.push_back({a[lane], lane < m ? vnext : e});
		}
		for (auto& w : wr) H[w.first] = w.second;
	}

	// WaveHeap7::push4: up to four pushes of ONE priority with one gather and one scatter. std::push_heap only touches the ancestors of the
	// new last element, so appending all k entries first and then sifting them up in order is the same sequence of heap states. Lane
	// (g, t) = 16 g + t keeps the CURRENT value of node a(g,t) = ((n + g + 1) >> t) - 1, the level-t ancestor of entry g's slot (t = 0: the
	// slot itself), through the k sift-ups: sift-up p moves the values on ITS path down by one as far as level m_p (the first ancestor whose
	// current priority is not greater stops it) and puts e_p there; every lane whose node lies on that path -- in whatever group -- takes its
	// new value from the lane above it in its own group (which tracks the same parent). Needs n >= 4 (no entry's ancestor is another new
	// entry) and n + k <= 32767 (an entry has at most 14 ancestors): the caller pushes one by one otherwise.
	void push4(const uint32_t* e, int k)
	{
		const uint32_t ed = e[0] >> 25;
		int a[64], dc[64]; uint32_t cur[64], orig[64]; bool on[64];
		for (int lane = 0; lane < 64; ++lane) {
			const int g = lane >> 4, t = lane & 15;
			const unsigned pos1 = (unsigned)(n + g) + 1u;
			const int depth = 31 - clz32(pos1);
			on[lane] = g < k && t <= depth;
			a[lane] = (int)(pos1 >> t) - 1;
			dc[lane] = depth - t;
			cur[lane] = t == 0 ? e[g < k ? g : 0] : get(a[lane], on[lane]);
			orig[lane] = cur[lane];
		}
		for (int p = 0; p < k; ++p) {
			const unsigned pos1 = (unsigned)(n + p) + 1u;
			const int depth = 31 - clz32(pos1);
			uint64_t stop = 0;
			for (int lane = 16 * p + 1; lane < 16 * p + 16; ++lane)
				if (on[lane] && (cur[lane] >> 25) <= ed) stop |= 1ull << (lane - 16 * p);
			const int m = stop ? __builtin_ctzll(stop) - 1 : depth;
			uint32_t nxt[64];
			for (int lane = 0; lane < 64; ++lane) {
				const uint32_t up = (lane & 15) < 15 ? cur[lane + 1] : 0u;        // row_shl:1 (the lane above in the same group)
				const int kk = depth - dc[lane];                                  // the level of path p at this node's depth
				const bool onpath = on[lane] && kk >= 0 && kk <= m && (int)(pos1 >> kk) - 1 == a[lane];
				nxt[lane] = onpath ? (kk < m ? up : e[p]) : cur[lane];
			}
			std::memcpy(cur, nxt, sizeof cur);
		}
		for (int lane = 0; lane < 64; ++lane)
			if (on[lane] && ((lane & 15) == 0 || cur[lane] != orig[lane])) H[a[lane]] = cur[lane];
		n += k;
	}

	// WaveHeap7::pop7 (the default since round 3): six levels per round, restated so that every node on the hole's path writes ONE slot
	// -- its own: H'[c_k] = H[c_{k+1}] for k < j (the value of the child the path goes to, which the lane has fetched anyway), H'[c_j] = value --
	// and so that the node with a left child only (stl_heap.h __adjust_heap's `(len & 1) == 0 && secondChild == (len - 2) / 2` case) is just a
	// node that "prefers" its only child: no tail case, no parent writes across blocks, the root (the hole) is an ordinary path node.
	uint32_t pop7()
	{
		const int len = n - 1;
		if (len == 0) { n = 0; return 0u; }
		const uint32_t value = H[len];
		const uint32_t vprio = value >> 25;
		uint32_t bchild[3][64]; int bg[3][64]; bool bp[3][64];
		std::memset(bchild, 0, sizeof bchild); std::memset(bg, 0, sizeof bg); std::memset(bp, 0, sizeof bp);
		int r0 = 0, j = 0;
		bool cont = true;
		uint32_t rootchild = 0;
		for (int B = 0; B < 3 && cont; ++B) {
			int gi[64]; uint32_t v[64], vl[64], vr[64];
			uint64_t right = 0, contm = 0;
			for (int lane = 0; lane < 64; ++lane) {
				const int d = 31 - clz32((unsigned)lane + 1u);
				gi[lane] = lane < 63 ? (r0 << d) + lane : 0x40000000;
				const long cl = 2L * gi[lane] + 1;
				const bool hasL = cl < len, inner = cl + 1 < len;
				v[lane] = gi[lane] < (int)H.size() ? H[gi[lane]] : 0u;          // (unconditional reads; out-of-range ones are never used)
				vl[lane] = cl < (long)H.size() ? H[cl] : 0u;
				vr[lane] = cl + 1 < (long)H.size() ? H[cl + 1] : 0u;
				if (inner && (vr[lane] >> 25) <= (vl[lane] >> 25)) right |= 1ull << lane;
				if (hasL) contm |= 1ull << lane;
			}
			uint64_t pm = 0, cm = 0;
			for (int lane = 0; lane < 63; ++lane) {
				const bool onp = (((right ^ dirm[lane]) & ancm[lane]) | (~contm & ancm[lane])) == 0;
				if (onp) pm |= 1ull << lane;
				if (onp && (v[lane] >> 25) <= vprio) cm |= 1ull << lane;
				bg[B][lane] = gi[lane]; bp[B][lane] = onp;
				bchild[B][lane] = ((right >> lane) & 1ull) ? vr[lane] : vl[lane];
			}
			if (B == 0) rootchild = bchild[0][0];
			if (cm) j = 6 * B + (31 - clz32((unsigned)(63 - clz64(cm)) + 1u));
			const int last = 63 - clz64(pm);                               // pm is never empty: the block's root is on the path
			if (last >= 31 && ((contm >> last) & 1ull)) r0 = 2 * gi[last] + 1 + (int)((right >> last) & 1ull);
			else cont = false;
		}
		std::vector<std::pair<int, uint32_t>> wr;
		for (int B = 0; B < 3; ++B)
			for (int lane = 0; lane < 63; ++lane) {
				const int dp = 6 * B + (31 - clz32((unsigned)lane + 1u));
				if (bp[B][lane] && dp <= j) wr.push_back({bg[B][lane], dp < j ? bchild[B][lane] : value});
			}
		for (size_t a = 0; a < wr.size(); ++a)
			for (size_t b = a + 1; b < wr.size(); ++b)
				if (wr[a].first == wr[b].first) return 0xFFFFFFFFu;          // every slot at most once
		for (auto& w : wr) { if (w.first < 0 || w.first >= len) return 0xFFFFFFFFu; H[w.first] = w.second; }
		n = len;
		return j == 0 ? value : rootchild;
	}

	// LV = levels per round: 5 = WaveHeap::pop, 6 = a first six-level pop (no longer in the kernels: pop7 below replaced it). Returns what the kernel reports as the root afterwards (0 if empty;
	// 0xFFFFFFFF if two of its scattered writes disagree about one slot).
	uint32_t pop(int LV)
	{
		const int len = n - 1;
		if (len == 0) { n = 0; return 0u; }
		const uint32_t value = H[len];
		const uint32_t vprio = value >> 25;
		const int half = (len - 1) / 2;
		const int NB = LV == 6 ? 3 : 4;
		const uint32_t pfv1 = len > 1 ? H[1] : 0u, pfv2 = len > 2 ? H[2] : 0u;   // what the prefetch holds in lanes 1 and 2
		uint32_t bv[4][64]; int bg[4][64]; bool bp[4][64];
		std::memset(bv, 0, sizeof bv); std::memset(bg, 0, sizeof bg); std::memset(bp, 0, sizeof bp);
		int r0 = 0, E = 0, depth = 0, j = 0;      // r0 = S - 1 of the kernel
		bool cont = half > 0;
		uint64_t right0 = 0;
		for (int B = 0; B < NB; ++B) {
			if (LV == 6 ? !cont : !(r0 < half)) continue;
			int gi[64]; uint32_t v[64], vl[64], vr[64]; bool inner[64];
			uint64_t right = 0, innerm = 0;
			for (int lane = 0; lane < 64; ++lane) {
				const int d = 31 - clz32((unsigned)lane + 1u);
				gi[lane] = (r0 << d) + lane;
				inner[lane] = (LV == 6 ? lane < 63 : lane < 31) && gi[lane] < half;
				v[lane] = get(gi[lane], lane < 63 && gi[lane] < len);
				vl[lane] = get(2 * gi[lane] + 1, inner[lane]);
				vr[lane] = get(2 * gi[lane] + 2, inner[lane]);
				if (inner[lane] && !((vr[lane] >> 25) > (vl[lane] >> 25))) right |= 1ull << lane;
				if (inner[lane]) innerm |= 1ull << lane;
			}
			if (B == 0) right0 = right;
			uint64_t pm = 0, cm = 0;
			for (int lane = 0; lane < 64; ++lane) {
				const bool onp = (ancm[lane] != 0 && ((right ^ dirm[lane]) & ancm[lane]) == 0 && (innerm & ancm[lane]) == ancm[lane]) || (LV == 6 && B > 0 && lane == 0);
				if (onp) pm |= 1ull << lane;
				if (onp && (v[lane] >> 25) <= vprio) cm |= 1ull << lane;
				bv[B][lane] = v[lane]; bg[B][lane] = gi[lane]; bp[B][lane] = onp;
			}
			if (LV == 6) {
				const int last = pm ? 63 - clz64(pm) : 0;
				const int gl = gi[last];
				depth = 6 * B + (31 - clz32((unsigned)last + 1u));
				if (cm) j = 6 * B + (31 - clz32((unsigned)(63 - clz64(cm)) + 1u));
				E = gl;
				if (last >= 31 && ((innerm >> last) & 1ull)) r0 = 2 * gl + 1 + (int)((right >> last) & 1ull);
				else cont = false;
			} else {
				if (pm) {
					const int last = 63 - clz64(pm);
					r0 = gi[last];
					depth += __builtin_popcountll(pm);
				}
				if (cm) j = 5 * B + (31 - clz32((unsigned)(63 - clz64(cm)) + 1u));
			}
		}
		if (LV == 5) E = r0;
		int tail = -1; uint32_t tailv = 0;
		if ((LV == 5 || !cont) && (len & 1) == 0 && E == (len - 2) / 2) {
			tail = 2 * E + 1;
			tailv = H[tail];
			++depth;
			if ((tailv >> 25) <= vprio) j = depth;
		}
		std::vector<std::pair<int, uint32_t>> wr;
		for (int B = 0; B < NB; ++B) {
			if (!(B == 0 || (LV == 6 ? 6 * B <= depth : 5 * B < depth))) continue;
			for (int lane = 0; lane < 64; ++lane) {
				const int d = 31 - clz32((unsigned)lane + 1u);
				const int dp = LV * B + d;
				if (bp[B][lane] && dp <= j) wr.push_back({(bg[B][lane] - 1) >> 1, bv[B][lane]});
				if (bp[B][lane] && dp == j) wr.push_back({bg[B][lane], value});
			}
		}
		if (tail >= 0 && j == depth) { wr.push_back({(tail - 1) >> 1, tailv}); wr.push_back({tail, value}); }
		if (j == 0) wr.push_back({0, value});
		// (every read above happened before any write, as in the kernel) no two writes may disagree about a slot
		for (size_t a = 0; a < wr.size(); ++a)
			for (size_t b = a + 1; b < wr.size(); ++b)
				if (wr[a].first == wr[b].first && wr[a].second != wr[b].second) return 0xFFFFFFFFu;
		for (auto& w : wr) { if (w.first < 0) return 0xFFFFFFFFu; H[w.first] = w.second; }
		n = len;
		if (LV == 5) return H[0];                                   // k_flood3<.., false> reads the root back
		const uint32_t c1 = half > 0 ? ((right0 & 1ull) ? pfv2 : pfv1) : tailv;
		return j == 0 ? value : c1;
	}
};


}  // namespace

// Random operation sequences shaped like the flood's (bursts of pushes with ONE priority, then a pop; priorities from a small alphabet so that
// ties are everywhere). After every operation the model's array equals what std::push_heap / std::pop_heap made of the same operations, and
// after every burst the entry on top equals k_flood3's prediction from (root after the pop, the burst). Returns 0, or the step that failed.
extern "C" long heap_model_fuzz(int LV, unsigned seed, int steps, int max_prio, int grow_bias, int hover, long* max_size)
{
	std::mt19937 rng(seed);
	WaveHeapModel M;
	std::vector<uint32_t> ref;
	uint32_t serial = 0;
	auto entry = [&](uint32_t prio) { return (prio << 25) | ((serial++ & 0x1FFFFFFu)); };   // distinct payloads: a wrong element in the right place is seen
	for (int s = 0; s < 8; ++s) { const uint32_t e = entry(s < 4 ? 0 : 1); M.push(e); ref.push_back(e); std::push_heap(ref.begin(), ref.end(), Cmp()); }
	for (long step = 1; step <= steps; ++step) {
		if (ref.empty()) break;
		// pop
		std::pop_heap(ref.begin(), ref.end(), Cmp());
		ref.pop_back();
		const uint32_t r = LV >= 7 ? M.pop7() : M.pop(LV);
		if (r == 0xFFFFFFFFu) return -step;
		if (M.n != (int)ref.size()) return step;
		if (!ref.empty() && (std::memcmp(M.H.data(), ref.data(), ref.size() * 4) != 0 || r != ref[0])) return step;
		// a burst of pushes with one priority
		const uint32_t root = ref.empty() ? 0u : ref[0];
		const int left = (int)ref.size();
		// hover > 0: the heap keeps returning to about `hover` entries (and runs empty now and then when hover is small)
		int burst = hover > 0 ? ((int)ref.size() < hover ? (int)(rng() % 13) : (int)(rng() % 2))
		                      : ((int)(rng() % 100) < grow_bias ? (int)(rng() % 13) : (int)(rng() % 3));
		if ((int)ref.size() + burst > (1 << 17)) burst = 0;
		if (max_size && (long)ref.size() + burst > *max_size) *max_size = (long)ref.size() + burst;
		const uint32_t ed = rng() % (uint32_t)max_prio;
		uint32_t first = 0;
		for (int q = 0; q < burst;) {
			// LV 8: pop7 and the pushes four at a time (push4) where its preconditions hold
			const int k = (LV == 8 && M.n >= 4 && M.n + 4 <= 32767) ? std::min(4, burst - q) : 1;
			uint32_t e[4];
			for (int i = 0; i < k; ++i) {
				e[i] = entry(ed);
				if (q + i == 0) first = e[i];
				ref.push_back(e[i]);
				std::push_heap(ref.begin(), ref.end(), Cmp());
			}
			if (LV == 8 && k > 1) M.push4(e, k);
			else if (LV == 8 && M.n >= 4 && M.n + 4 <= 32767 && (rng() & 1)) M.push4(e, 1);
			else M.push(e[0]);
			q += k;
			if (M.n != (int)ref.size() || std::memcmp(M.H.data(), ref.data(), ref.size() * 4) != 0) return step;
		}
		if (!ref.empty()) {
			const uint32_t predicted = (burst > 0 && (left == 0 || ed < (root >> 25))) ? first : root;
			if (predicted != ref[0]) return step;
		}
	}
	return 0;
}
