/* oracle/cimbar_oracle_extract.c -- CPU restatement of the stage in FRONT of the frame decoder (BASELINE config 5):
 * the image preparation of Scanner (src/lib/extractor/Scanner.h:148-165) and Deskewer::deskew (src/lib/extractor/Deskewer.h:26-40).
 * The anchor search itself (Scanner::scan, Scanner.h:277-405 / Scanner.cpp) is host logic on the thresholded image and is not restated:
 * tests take the corners from the reference build.
 *
 * TEST INFRASTRUCTURE, like cimbar_oracle.c. PARITY UNPINNED at the OpenCV boundary: the reference calls cv::cvtColor, cv::GaussianBlur,
 * cv::threshold(OTSU), cv::getPerspectiveTransform and cv::warpPerspective, OpenCV is not in /root/reference, and the reference's tests
 * for this stage need its samples/ images. What is pinned: this file == the reference's own Scanner / Deskewer / Extractor code
 * compiled against oracle/cvshim (tests/test_oracle_vs_ref.py), i.e. the restatement of OpenCV 4.5.x's published arithmetic is
 * the same on both sides [assumed-OpenCV].
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cimbar_oracle.h"

/* Scanner::nextPowerOfTwoPlusOne, Scanner.h:92-103 */
static unsigned next_pow2_plus_one(unsigned v)
{
	v--;
	v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
	return v + 2 > 3u ? v + 2 : 3u;
}

static int refl101(int i, int n)
{
	if (n == 1) return 0;
	while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
	return i;
}

/* Scanner::preprocess_image(img, fast = true), Scanner.h:148-165 -> threshold_fast :126-130.
 *   cvtColor(RGB2GRAY)                 (R*9798 + G*19235 + B*3735 + 2^14) >> 15                                  [assumed-OpenCV]
 *   GaussianBlur(unit x unit, sigma 0) unit = max(3, nextPow2(min(w,h) * 0.002) + 1): 3 up to 1500 px, 5 above; small fixed kernels
 *                                      [1 2 1]/4, [1 4 6 4 1]/16, fixed point, one rounding, BORDER_REFLECT_101             [assumed-OpenCV]
 *   threshold(0, 255, BINARY | OTSU)   getThreshVal_Otsu_8u on the blurred image, out = v > t ? 255 : 0                 [assumed-OpenCV]
 * out: w*h bytes. Returns the Otsu threshold (or -1 for an unsupported kernel size). */
/* the first two steps alone -- cvtColor(RGB2GRAY) + GaussianBlur(unit x unit, 0) -- so that the blur can be pinned by itself (tests/test_opencv_pin_vectors.py).
 * out: w*h bytes; hist (may be NULL): the blurred image's histogram. Returns the kernel size, or -1 for an unsupported one. */
int co_gray_blur(const uint8_t* rgb, int w, int h, uint8_t* out, int* hist_out)
{
	const size_t n = (size_t)w * h;
	unsigned unit = (unsigned)(w < h ? w : h);
	unit = next_pow2_plus_one((unsigned)(unit * 0.002));
	if (unit != 3 && unit != 5 && unit != 9 && unit != 17) return -1;   /* (33 and up: captures of 8500 px and more on the short side) */
	/* ksize 9: getGaussianKernel(9, sigma <= 0) -> sigma = 0.3*((9-1)*0.5 - 1) + 0.8 = 1.7, exp(-x^2 / (2 sigma^2)) normalised, then the 8.8 fixed-point
 * weights of getGaussianKernelFixedPoint_ED (round outside-in with the error carried, centre = 256 - the rest): 256 * k = 3.80 12.75 30.29 50.90
 * 60.51 -> {4, 13, 30, 51, 60, 51, 30, 13, 4} (plain rounding gives the same, no value is near a half), s = 16 [assumed-OpenCV] */
	/* ksize 17: sigma = 0.3*((17-1)*0.5 - 1) + 0.8 = 2.9; 256 * k = 0.786 1.919 4.156 7.992 13.647 20.691 27.853 33.292 35.331; rounded outside-in with the
	 * error carried: 1 2 4 8 13 21 28 33, centre 256 - 2 * 110 = 36 (the fifth weight is 13.4993 before rounding) [assumed-OpenCV] */
	const int r = (int)unit / 2, shift = r == 1 ? 4 : (r == 2 ? 8 : 16);
	static const int k3[3] = {1, 2, 1}, k5[5] = {1, 4, 6, 4, 1}, k9[9] = {4, 13, 30, 51, 60, 51, 30, 13, 4},
	                 k17[17] = {1, 2, 4, 8, 13, 21, 28, 33, 36, 33, 28, 21, 13, 8, 4, 2, 1};
	const int* k = r == 1 ? k3 : (r == 2 ? k5 : (r == 4 ? k9 : k17));

	uint8_t* gray = (uint8_t*)malloc(n);
	int* hs = (int*)malloc(n * sizeof(int));
	for (size_t i = 0; i < n; ++i)
		gray[i] = (uint8_t)((rgb[3 * i] * 9798 + rgb[3 * i + 1] * 19235 + rgb[3 * i + 2] * 3735 + (1 << 14)) >> 15);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			int acc = 0;
			for (int t = -r; t <= r; ++t) acc += k[t + r] * gray[(size_t)y * w + refl101(x + t, w)];
			hs[(size_t)y * w + x] = acc;
		}
	int hist[256];
	memset(hist, 0, sizeof hist);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			int acc = 0;
			for (int t = -r; t <= r; ++t) acc += k[t + r] * hs[(size_t)refl101(y + t, h) * w + x];
			const uint8_t v = (uint8_t)((acc + (1 << (shift - 1))) >> shift);
			out[(size_t)y * w + x] = v;
			hist[v]++;
		}
	free(hs); free(gray);
	if (hist_out) memcpy(hist_out, hist, sizeof hist);
	return (int)unit;
}

int co_scan_preprocess(const uint8_t* rgb, int w, int h, uint8_t* out)
{
	const size_t n = (size_t)w * h;
	int hist[256];
	if (co_gray_blur(rgb, w, h, out, hist) < 0) return -1;


	/* thresh.cpp getThreshVal_Otsu_8u, double arithmetic in this order */
	double mu = 0, scale = 1. / ((double)w * h);
	for (int i = 0; i < 256; ++i) mu += i * (double)hist[i];
	mu *= scale;
	double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
	for (int i = 0; i < 256; ++i) {
		double p_i, q2, mu2, sigma;
		p_i = hist[i] * scale;
		mu1 *= q1;
		q1 += p_i;
		q2 = 1. - q1;
		if ((q1 < q2 ? q1 : q2) < FLT_EPSILON || (q1 > q2 ? q1 : q2) > 1. - FLT_EPSILON) continue;
		mu1 = (mu1 + i * p_i) / q1;
		mu2 = (mu - q1 * mu1) / q2;
		sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
		if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
	}
	const int t = (int)max_val;
	for (size_t i = 0; i < n; ++i) out[i] = out[i] > t ? 255 : 0;
	return t;
}

/* Deskewer::deskew's matrix, Deskewer.h:28-36: corners (top-left, top-right, bottom-left, bottom-right, as the float pairs Corners::all()
 * returns, Corners.h:45-53) -> (anchor, anchor), (size-anchor, anchor), (anchor, size-anchor), (size-anchor, size-anchor) with
 * size = 1024, anchor = 30, padding 0; cv::getPerspectiveTransform = 8x8 system solved by LU with partial pivoting  [assumed-OpenCV].
 * m9: row-major 3x3, m9[8] = 1. Returns 1, or 0 for a singular system (all zeros then). */
int co_perspective_transform(const float* src8, const float* dst8, double* m9)
{
	double a[8][8], b[8];
	for (int i = 0; i < 4; ++i) {
		const float sx = src8[2 * i], sy = src8[2 * i + 1], dx = dst8[2 * i], dy = dst8[2 * i + 1];
		a[i][0] = a[i + 4][3] = sx;
		a[i][1] = a[i + 4][4] = sy;
		a[i][2] = a[i + 4][5] = 1;
		a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
		a[i][6] = -sx * dx;      /* float products, then widened: Point2f arithmetic in the reference's call */
		a[i][7] = -sy * dx;
		a[i + 4][6] = -sx * dy;
		a[i + 4][7] = -sy * dy;
		b[i] = dx;
		b[i + 4] = dy;
	}
	double* A = &a[0][0];
	const int m = 8;
	const double eps = DBL_EPSILON * 100;
	int ok = 1;
	for (int i = 0; i < m && ok; ++i) {
		int k = i;
		for (int j = i + 1; j < m; ++j) if (fabs(A[j * m + i]) > fabs(A[k * m + i])) k = j;
		if (fabs(A[k * m + i]) < eps) { ok = 0; break; }
		if (k != i) {
			for (int j = i; j < m; ++j) { double t = A[i * m + j]; A[i * m + j] = A[k * m + j]; A[k * m + j] = t; }
			double t = b[i]; b[i] = b[k]; b[k] = t;
		}
		const double d = -1 / A[i * m + i];
		for (int j = i + 1; j < m; ++j) {
			const double alpha = A[j * m + i] * d;
			for (int c = i + 1; c < m; ++c) A[j * m + c] += alpha * A[i * m + c];
			b[j] += alpha * b[i];
		}
	}
	if (ok)
		for (int i = m - 1; i >= 0; --i) {
			double s = b[i];
			for (int c = i + 1; c < m; ++c) s -= A[i * m + c] * b[c];
			b[i] = s / A[i * m + i];
		}
	for (int i = 0; i < 8; ++i) m9[i] = ok ? b[i] : 0;
	m9[8] = 1.;
	return ok;
}

void co_deskew_points(float* dst8)
{
	const float sx = CO_IMG_W, sy = CO_IMG_H, anchor = 30;   /* Config::image_size_x/y(), Config::anchor_size(); padding 0 (Deskewer.h:28-32) */
	dst8[0] = anchor; dst8[1] = anchor;
	dst8[2] = sx - anchor; dst8[3] = anchor;
	dst8[4] = anchor; dst8[5] = sy - anchor;
	dst8[6] = sx - anchor; dst8[7] = sy - anchor;
}

/* cv::warpPerspective(img, output, transform, output.size(), INTER_LINEAR), Deskewer.h:38  [assumed-OpenCV imgwarp.cpp]:
 * transform inverted (3x3 closed form); per 64x16 destination block the source position of every pixel in 1/32-pixel fixed point
 * (X0 = M0*x + M1*(y+y1) + M2 at the block's left edge, fX = (X0 + M0*x1) * (32 / (W0 + M6*x1)), X = cvRound(fX)); fixed-point
 * bilinear weights (32-fx)(32-fy)*32 .. summing to 2^15, out = (sum + 2^14) >> 15; taps outside the source read 0 (BORDER_CONSTANT). */
int co_warp_perspective(const uint8_t* rgb, int sw, int sh, const double* m9, uint8_t* out, int width, int height)
{
	const double* S = m9;
	double M[9];
	double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
	if (d != 0.) {
		d = 1. / d;
		M[0] = (S[4] * S[8] - S[5] * S[7]) * d; M[1] = (S[2] * S[7] - S[1] * S[8]) * d; M[2] = (S[1] * S[5] - S[2] * S[4]) * d;
		M[3] = (S[5] * S[6] - S[3] * S[8]) * d; M[4] = (S[0] * S[8] - S[2] * S[6]) * d; M[5] = (S[2] * S[3] - S[0] * S[5]) * d;
		M[6] = (S[3] * S[7] - S[4] * S[6]) * d; M[7] = (S[1] * S[6] - S[0] * S[7]) * d; M[8] = (S[0] * S[4] - S[1] * S[3]) * d;
	} else {
		for (int i = 0; i < 9; ++i) M[i] = 0;
	}
	const int BLOCK_SZ = 32;
	int bh0 = BLOCK_SZ / 2 < height ? BLOCK_SZ / 2 : height;
	int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < width ? BLOCK_SZ * BLOCK_SZ / bh0 : width;
	bh0 = BLOCK_SZ * BLOCK_SZ / bw0 < height ? BLOCK_SZ * BLOCK_SZ / bw0 : height;
	for (int y = 0; y < height; y += bh0)
		for (int x = 0; x < width; x += bw0) {
			const int bw = bw0 < width - x ? bw0 : width - x, bh = bh0 < height - y ? bh0 : height - y;
			for (int y1 = 0; y1 < bh; ++y1) {
				const double X0 = M[0] * x + M[1] * (y + y1) + M[2];
				const double Y0 = M[3] * x + M[4] * (y + y1) + M[5];
				const double W0 = M[6] * x + M[7] * (y + y1) + M[8];
				uint8_t* o = out + ((size_t)(y + y1) * width + x) * 3;
				for (int x1 = 0; x1 < bw; ++x1) {
					double W = W0 + M[6] * x1;
					W = W ? 32. / W : 0;
					double fX = (X0 + M[0] * x1) * W, fY = (Y0 + M[3] * x1) * W;
					fX = fX < (double)INT_MAX ? fX : (double)INT_MAX; fX = fX > (double)INT_MIN ? fX : (double)INT_MIN;
					fY = fY < (double)INT_MAX ? fY : (double)INT_MAX; fY = fY > (double)INT_MIN ? fY : (double)INT_MIN;
					const int X = (int)lrint(fX), Y = (int)lrint(fY);
					int sx = X >> 5, sy = Y >> 5;
					sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
					sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
					const int fx = X & 31, fy = Y & 31;
					const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
					for (int c = 0; c < 3; ++c) {
#define PX(yy, xx) (((xx) < 0 || (xx) >= sw || (yy) < 0 || (yy) >= sh) ? 0 : (int)rgb[((size_t)(yy) * sw + (xx)) * 3 + c])
						const int v = PX(sy, sx) * w00 + PX(sy, sx + 1) * w01 + PX(sy + 1, sx) * w10 + PX(sy + 1, sx + 1) * w11;
#undef PX
						o[x1 * 3 + c] = (uint8_t)((v + (1 << 14)) >> 15);
					}
				}
			}
		}
	return 0;
}

/* Deskewer::deskew (image_size_x x image_size_y output -- 1024x1024 in mode B --, anchor 30, padding 0) from the corners Corners::all() would return */
int co_deskew(const uint8_t* rgb, int sw, int sh, const float* corners8, uint8_t* out1024)
{
	float dst8[8];
	double m9[9];
	co_deskew_points(dst8);
	co_perspective_transform(corners8, dst8, m9);
	return co_warp_perspective(rgb, sw, sh, m9, out1024, CO_IMG_W, CO_IMG_H);
}

/* ------------------------------------------------------------------------------------------------ anchor search
 * Scanner::scan (src/lib/extractor/Scanner.cpp:183-202) on the 0/255 image of co_scan_preprocess, restated:
 *   ScanState                       ScanState.h:21-104   run-length state machine looking for active:inactive runs 1:1:4:1:1 (limits 3..6,
 *                                                        "114") resp. 1:2:2:2:1 ("122", the smaller bottom-right anchor)
 *   scan_horizontal/vertical/diagonal Scanner.h:167-275
 *   t1 rows -> t2 column -> t3 diagonal -> t4 confirm    Scanner.h:277-405
 *   deduplication while scanning (on_t1_scan)            Scanner.h:391-405, Anchor::is_mergeable Anchor.h:87-95
 *   filter_candidates / sort_top_to_bottom / add_bottom_right_corner   Scanner.cpp:79-181
 * dark mode only (test_pixel: pixel > 127, Scanner.cpp:52-59). */
typedef struct { int x, xmax, y, ymax; } anchor_t;

static int a_xavg(const anchor_t* a) { return (a->x + a->xmax) / 2; }
static int a_yavg(const anchor_t* a) { return (a->y + a->ymax) / 2; }
static int a_xrange(const anchor_t* a) { return abs(a->x - a->xmax) / 2; }
static int a_yrange(const anchor_t* a) { return abs(a->y - a->ymax) / 2; }
static int a_max_range(const anchor_t* a) { int p = abs(a->x - a->xmax), q = abs(a->y - a->ymax); return p > q ? p : q; }
static unsigned long long a_size(const anchor_t* a)
{
	return (unsigned long long)(pow((double)(a->x - a->xmax), 2) + pow((double)(a->y - a->ymax), 2));   /* Anchor.h:77-80 */
}
static void a_merge(anchor_t* a, const anchor_t* o)
{
	if (o->x < a->x) a->x = o->x;
	if (o->xmax > a->xmax) a->xmax = o->xmax;
	if (o->y < a->y) a->y = o->y;
	if (o->ymax > a->ymax) a->ymax = o->ymax;
}
/* Anchor.h:87-95 (a division by a zero max_range cannot happen for anchors the scans produce: every run pattern is >= 5 px long) */
static int a_mergeable(const anchor_t* a, const anchor_t* rhs, int max_distance)
{
	if (abs(a_xavg(a) - a_xavg(rhs)) > max_distance || abs(a_yavg(a) - a_yavg(rhs)) > max_distance) return 0;
	const int mr = a_max_range(a);
	if (mr == 0) return 0;
	const int ratio = a_max_range(rhs) * 10 / mr;
	return ratio > 6 && ratio < 17;
}

typedef struct { int state; int tally[8]; int nt; const float (*limits)[2]; } scan_state;
static const float LIM_114[6][2] = {{0, 0}, {3.0f, 6.0f}, {3.0f, 6.0f}, {0, 0}, {3.0f, 6.0f}, {3.0f, 6.0f}};
static const float LIM_122[6][2] = {{0, 0}, {1.0f, 3.0f}, {0.5f, 1.5f}, {0, 0}, {0.5f, 1.5f}, {1.0f, 3.0f}};
static void ss_init(scan_state* s, int kind) { s->state = 0; s->nt = 1; s->tally[0] = 0; s->limits = kind == 114 ? LIM_114 : LIM_122; }
/* ScanState::process, ScanState.h:21-60 (+ evaluate_state :70-97, pop_state :63-68) */
static int ss_process(scan_state* s, int active)
{
	const int even = s->state == 0 || s->state == 2 || s->state == 4;
	const int odd = s->state == 1 || s->state == 3 || s->state == 5;
	if ((even && active) || (odd && !active)) {
		s->state += 1;
		s->tally[s->nt++] = 1;
		if (s->state == 6) {
			int res = -1, okp = 1;
			for (int i = 1; i <= 5; ++i) if (s->tally[i] == 0) okp = 0;
			if (okp) {
				const float center = (float)s->tally[3];
				for (int i = 1; i <= 5 && okp; ++i) {
					if (i == 3) continue;
					const float ratio_min = center / (float)(s->tally[i] + 1);
					const float ratio_max = center / (float)(s->tally[i] - 1 > 1 ? s->tally[i] - 1 : 1);
					if (ratio_max < s->limits[i][0] || ratio_min > s->limits[i][1]) okp = 0;
				}
				if (okp) { res = 0; for (int i = 1; i <= 5; ++i) res += s->tally[i]; }
			}
			s->state -= 2;
			for (int i = 0; i + 2 < s->nt; ++i) s->tally[i] = s->tally[i + 2];
			s->nt -= 2;
			return res;
		}
		return -1;
	}
	if (odd && active) s->tally[s->nt - 1] += 1;
	if (!active && (s->state == 2 || s->state == 4)) s->tally[s->nt - 1] += 1;
	return -1;
}

typedef struct { const uint8_t* img; int w, h, skip, cutoff; } scanner_t;
typedef struct { anchor_t* v; int n, cap; } alist;
static __thread int g_dbg_max_list = 0;   /* longest list any scan built during the last co_scan_anchors (sizes the device kernel's fixed lists) */
int co_scan_debug_max_list(void) { return g_dbg_max_list; }
static void al_push(alist* l, anchor_t a)
{
	if (l->n + 1 > g_dbg_max_list) g_dbg_max_list = l->n + 1;
	if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 64; l->v = (anchor_t*)realloc(l->v, sizeof(anchor_t) * (size_t)l->cap); }
	l->v[l->n++] = a;
}
static int test_pixel(const scanner_t* sc, int x, int y)
{
	if (x < 0 || y < 0 || x >= sc->w || y >= sc->h) return 0;   /* the reference reads out of bounds here (confirm scans next to the image border) */
	return sc->img[(size_t)y * sc->w + x] > 127;
}

/* Scanner.h:167-197 */
static int scan_horizontal(const scanner_t* sc, int kind, alist* pts, int y, int xstart, int xend)
{
	if (xstart < 0) xstart = 0;
	if (xend < 0 || xend > sc->w) xend = sc->w;
	const int init = pts->n;
	scan_state st;
	ss_init(&st, kind);
	for (int x = xstart; x < xend; ++x) {
		const int res = ss_process(&st, test_pixel(sc, x, y));
		if (res > 0) { anchor_t a = {x - res, x - 1, y, y}; al_push(pts, a); }
	}
	const int res = ss_process(&st, 0);
	if (res > 0) { anchor_t a = {xend - res, xend - 1, y, y}; al_push(pts, a); }
	return init != pts->n;
}
/* Scanner.h:199-233 */
static int scan_vertical(const scanner_t* sc, int kind, alist* pts, int x, int xmax, int ystart, int yend)
{
	if (xmax < 0) xmax = x;
	const int xavg = (x + xmax) / 2;
	if (ystart < 0) ystart = 0;
	if (yend < 0 || yend > sc->h) yend = sc->h;
	const int init = pts->n;
	scan_state st;
	ss_init(&st, kind);
	for (int y = ystart; y < yend; ++y) {
		const int res = ss_process(&st, test_pixel(sc, xavg, y));
		if (res > 0) { anchor_t a = {xavg, xavg, y - res, y - 1}; al_push(pts, a); }
	}
	const int res = ss_process(&st, 0);
	if (res > 0) { anchor_t a = {xavg, xavg, yend - res, yend - 1}; al_push(pts, a); }
	return init != pts->n;
}
/* Scanner.h:235-275 */
static int scan_diagonal(const scanner_t* sc, int kind, alist* pts, int xstart, int xend, int ystart, int yend)
{
	if (xend > sc->w) xend = sc->w;
	if (yend > sc->h) yend = sc->h;
	if (xstart < 0) { const int off = -xstart; xstart += off; ystart += off; }
	if (ystart < 0) { const int off = -ystart; xstart += off; ystart += off; }
	const int init = pts->n;
	scan_state st;
	ss_init(&st, kind);
	int x = xstart, y = ystart;
	for (; x < xend && y < yend; ++x, ++y) {
		const int res = ss_process(&st, test_pixel(sc, x, y));
		if (res > 0) { anchor_t a = {x - res, x - 1, y - res, y - 1}; al_push(pts, a); }
	}
	const int res = ss_process(&st, 0);
	if (res > 0) { anchor_t a = {x - res, x - 1, y - res, y - 1}; al_push(pts, a); }
	return init != pts->n;
}

/* t4_confirm_scan, Scanner.h:341-389: returns 1 and the (possibly grown) anchor in *hint */
static int t4_confirm(const scanner_t* sc, int kind, anchor_t* hint, int merge_confirms)
{
	{
		alist cf = {0, 0, 0};
		const int xstart = hint->x - a_xrange(hint), xend = hint->xmax + a_xrange(hint), yavg = a_yavg(hint);
		for (int dy = -1; dy <= 1; ++dy)
			if (!scan_horizontal(sc, kind, &cf, yavg + dy, xstart, xend)) { free(cf.v); return 0; }
		int confirm = 0;
		for (int k = 0; k < cf.n; ++k)
			if (a_mergeable(&cf.v[k], hint, sc->cutoff)) {
				confirm = 1;
				if (!merge_confirms) break;
				a_merge(hint, &cf.v[k]);
			}
		free(cf.v);
		if (!confirm) return 0;
	}
	{
		alist cf = {0, 0, 0};
		const int ystart = hint->y - a_yrange(hint), yend = hint->ymax + a_yrange(hint), xavg = a_xavg(hint);
		for (int dx = -1; dx <= 1; ++dx)
			if (!scan_vertical(sc, kind, &cf, xavg + dx, xavg + dx, ystart, yend)) { free(cf.v); return 0; }
		int confirm = 0;
		for (int k = 0; k < cf.n; ++k)
			if (a_mergeable(&cf.v[k], hint, sc->cutoff)) {
				confirm = 1;
				if (!merge_confirms) break;
				a_merge(hint, &cf.v[k]);
			}
		free(cf.v);
		if (!confirm) return 0;
	}
	return 1;
}

/* on_t1_scan, Scanner.h:391-405: t2 column (:295-306) -> t3 diagonal (:308-339) -> t4 confirm */
static void on_t1_scan(const scanner_t* sc, int kind, const anchor_t* found, alist* candidates, int merge_confirms)
{
	for (int k = 0; k < candidates->n; ++k)
		if (a_mergeable(&candidates->v[k], found, sc->cutoff)) return;
	alist col = {0, 0, 0};
	scan_vertical(sc, kind, &col, found->x, found->xmax, found->y - 3 * a_xrange(found), found->ymax + 3 * a_xrange(found));
	for (int k = 0; k < col.n; ++k) {
		const anchor_t* p = &col.v[k];
		alist dg = {0, 0, 0};
		const int yr = a_yrange(p);
		if (scan_diagonal(sc, kind, &dg, a_xavg(p) - 2 * yr, a_xavg(p) + 2 * yr, p->y - yr, p->ymax + yr)) {
			int confirm = 0;
			anchor_t merged = *p;
			for (int q = 0; q < dg.n; ++q)
				if (a_mergeable(&dg.v[q], p, sc->cutoff)) { confirm = 1; a_merge(&merged, &dg.v[q]); }
			if (confirm && t4_confirm(sc, kind, &merged, merge_confirms)) al_push(candidates, merged);
		}
		free(dg.v);
	}
	free(col.v);
}

/* t1_scan_rows, Scanner.h:277-293 */
static void t1_scan_rows(const scanner_t* sc, int kind, alist* candidates, int merge_confirms, int skip, int y, int yend, int xstart, int xend)
{
	if (skip <= 0) skip = sc->skip;
	if (y < 0) y = skip;
	if (yend < 0 || yend > sc->h) yend = sc->h;
	alist pts = {0, 0, 0};
	for (; y < yend; y += skip) scan_horizontal(sc, kind, &pts, y, xstart, xend);
	for (int k = 0; k < pts.n; ++k) on_t1_scan(sc, kind, &pts.v[k], candidates, merge_confirms);
	free(pts.v);
}

/* Scanner::scan for a w x h image of 0 / 255. anchors: up to 4 x {x, xmax, y, ymax} in the reference's order (top-left, top-right,
 * bottom-left, bottom-right). Returns how many it found (Scanner.cpp:183-202). */
int co_scan_anchors(const uint8_t* binary, int w, int h, int32_t* anchors16)
{
	scanner_t sc = {binary, w, h, (h < w ? h : w) / 60, w / 30};   /* Scanner.h:168-174: _skip, _mergeCutoff */
	g_dbg_max_list = 0;
	alist cand = {0, 0, 0};
	t1_scan_rows(&sc, 114, &cand, 1, -1, -1, -1, -1, -1);            /* scan_primary, Scanner.cpp:171-181 */
	/* filter_candidates, Scanner.cpp:79-103 (std::sort on <= 16 elements is an insertion sort: stable) */
	unsigned cutoff = 0;
	if (cand.n >= 3) {
		for (int i = 1; i < cand.n; ++i) {
			anchor_t key = cand.v[i];
			int j = i - 1;
			while (j >= 0 && a_size(&key) > a_size(&cand.v[j])) { cand.v[j + 1] = cand.v[j]; --j; }
			cand.v[j + 1] = key;
		}
		unsigned long long cs = 0;
		for (int i = 0; i < 3; ++i) cs += a_size(&cand.v[i]);
		cutoff = (unsigned)cs;     /* `unsigned cutoff` accumulates the unsigned long long sizes */
		cutoff /= 8;
		int i = 0;
		for (; i < cand.n; ++i) if (a_size(&cand.v[i]) < cutoff) break;
		if (i > 3) i = 3;
		if (i < cand.n) cand.n = i;
	}
	/* sort_top_to_bottom, Scanner.cpp:105-139 */
	if (cand.n >= 3) {
		int cx[3], cy[3];
		for (int i = 0; i < 3; ++i) { cx[i] = a_xavg(&cand.v[i]); cy[i] = a_yavg(&cand.v[i]); }
		const int ex[3] = {cx[1] - cx[2], cx[2] - cx[0], cx[0] - cx[1]}, ey[3] = {cy[1] - cy[2], cy[2] - cy[0], cy[0] - cy[1]};
		int tl = 0, maxd = 0;
		for (int i = 0; i < 3; ++i) { const int d = ex[i] * ex[i] + ey[i] * ey[i]; if (d > maxd) { tl = i; maxd = d; } }
		const int dep = tl - 1 < 0 ? 2 : tl - 1, inc = tl + 1 >= 3 ? 0 : tl + 1;
		const int ix = -ey[inc], iy = ex[inc];
		const int ox = ex[dep] - ix, oy = ey[dep] - iy;
		int tr, bl;
		if (ox * ox + oy * oy < ex[dep] * ex[dep] + ey[dep] * ey[dep]) { tr = inc; bl = dep; }
		else { tr = dep; bl = inc; }
		anchor_t a0 = cand.v[tl], a1 = cand.v[tr], a2 = cand.v[bl];
		cand.v[0] = a0; cand.v[1] = a1; cand.v[2] = a2;
		cand.n = 3;
	}
	/* add_bottom_right_corner, Scanner.cpp:141-181 */
	if (cand.n == 3 && cutoff != 0) {
		const anchor_t* a = cand.v;
		const int mr0 = a_max_range(&a[0]), mr1 = a_max_range(&a[1]), mr2 = a_max_range(&a[2]);
		const double top_scalar = mr2 / (double)(mr1 > mr0 ? mr1 : mr0);
		const int tex = (int)((a_xavg(&a[1]) - a_xavg(&a[0])) * top_scalar), tey = (int)((a_yavg(&a[1]) - a_yavg(&a[0])) * top_scalar);
		const int g1x = a_xavg(&a[2]) + tex, g1y = a_yavg(&a[2]) + tey;
		const double left_scalar = mr1 / (double)(mr2 > mr0 ? mr2 : mr0);
		const int lex = (int)((a_xavg(&a[2]) - a_xavg(&a[0])) * left_scalar), ley = (int)((a_yavg(&a[2]) - a_yavg(&a[0])) * left_scalar);
		const int g2x = a_xavg(&a[1]) + lex, g2y = a_yavg(&a[1]) + ley;
		const int ccx = (g1x + g2x) / 2, ccy = (g1y + g2y) / 2;
		int mrmax = mr0 > mr1 ? mr0 : mr1;
		if (mr2 > mrmax) mrmax = mr2;
		const int range = (int)((float)mrmax * 2.0f);
		alist c2 = {0, 0, 0};
		t1_scan_rows(&sc, 122, &c2, 0, sc.skip / 2, ccy - range, ccy + range, ccx - range, ccx + range);
		for (int k = 0; k < c2.n; ++k)
			if (a_size(&c2.v[k]) > cutoff) { al_push(&cand, c2.v[k]); break; }
		free(c2.v);
	}
	const int n = cand.n < 4 ? cand.n : 4;
	for (int i = 0; i < n; ++i) { anchors16[4 * i] = cand.v[i].x; anchors16[4 * i + 1] = cand.v[i].xmax; anchors16[4 * i + 2] = cand.v[i].y; anchors16[4 * i + 3] = cand.v[i].ymax; }
	const int total = cand.n;
	free(cand.v);
	return total;
}

/* Extractor::extract (Extractor.h:29-45): 0 FAILURE, 1 SUCCESS, 2 NEEDS_SHARPEN; corners8 = Corners::all() (Corners.h:45-53), out = the
 * deskewed 1024x1024 RGB8 frame. */
int co_extract(const uint8_t* rgb, int w, int h, uint8_t* out1024, float* corners8)
{
	uint8_t* bin = (uint8_t*)malloc((size_t)w * h);
	if (co_scan_preprocess(rgb, w, h, bin) < 0) { free(bin); return 0; }
	int32_t an[16];
	const int found = co_scan_anchors(bin, w, h, an);
	free(bin);
	if (found < 4) return 0;
	int cx[4], cy[4];
	for (int i = 0; i < 4; ++i) { cx[i] = (an[4 * i] + an[4 * i + 1]) / 2; cy[i] = (an[4 * i + 2] + an[4 * i + 3]) / 2; }
	float c8[8];
	for (int i = 0; i < 4; ++i) { c8[2 * i] = (float)cx[i]; c8[2 * i + 1] = (float)cy[i]; }
	if (corners8) memcpy(corners8, c8, sizeof c8);
	co_deskew(rgb, w, h, c8, out1024);
	/* Corners::is_granular_scale({1024, 1024}), Corners.h:55-73: every edge longer than the output in x or in y, else we are upscaling */
	const int e[4][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}};   /* tl-tr, tr-br, br-bl, bl-tl */
	int granular = 1;
	for (int k = 0; k < 4; ++k) {
		const int a = e[k][0], b = e[k][1];
		if (!(abs(cx[a] - cx[b]) > CO_IMG_W || abs(cy[a] - cy[b]) > CO_IMG_H)) granular = 0;
	}
	return granular ? 1 : 2;
}

/* ---- the capture formats of the reference's C ABI: get_rgb (src/lib/cimbar_js/cimbar_recv_js.cpp:94-120) behind
 * cimbard_scan_extract_decode(img, w, h, format, ...) (cimbar_recv_js.h:17; callers web/recv-worker.js:38-47). `format` as the reference
 * reads it: 12 = NV12 -> cvtColor(COLOR_YUV2RGB_NV12); 420 -> cvtColor(COLOR_YUV420p2RGB), which OpenCV defines as COLOR_YUV2RGB_YV12 (the
 * plane behind Y is read as V, the next one as U -- kept, not "fixed"); 4 = RGBA -> cvtColor(COLOR_RGBA2RGB); everything else (3, and any
 * value get_rgb's `default:` lets through) = RGB8 as it is; `format <= 0` is 3 (:150-151).
 * [assumed-OpenCV] color_yuv.simd.hpp, BT.601 in 20-bit fixed point: ITUR_BT_601_CY 1220542, CUB 2116026, CUG -409993, CVG -852492, CVR 1673527;
 *   ruv = 2^19 + CVR (v - 128);  guv = 2^19 + CVG (v - 128) + CUG (u - 128);  buv = 2^19 + CUB (u - 128);  y' = max(0, y - 16) CY
 *   R, G, B = saturate_cast<uchar>((y' + cuv) >> 20)
 * One (u, v) pair per 2x2 block of pixels. The 4:2:0 layouts need an even width and an even height (OpenCV asserts that; the reference would
 * throw): co_capture_bytes returns 0 for them and co_capture_to_rgb -1. */
size_t co_capture_bytes(int w, int h, int format)
{
	if (w <= 0 || h <= 0) return 0;
	if (format == 12 || format == 420) return (w % 2 || h % 2) ? 0 : (size_t)w * h * 3 / 2;
	return (size_t)w * h * (format == 4 ? 4 : 3);
}

static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

int co_capture_to_rgb(const uint8_t* img, int w, int h, int format, uint8_t* rgb)
{
	if (format <= 0) format = 3;
	if (!co_capture_bytes(w, h, format)) return -1;
	const size_t n = (size_t)w * h;
	if (format == 12 || format == 420) {
		const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;
		const uint8_t* cplane = img + n;              /* NV12: h/2 rows of w bytes (u, v, u, v ...); 420: a w/2 x h/2 plane read as V, then one read as U */
		const size_t csize = (size_t)(w / 2) * (h / 2);
		for (int y = 0; y < h; ++y)
			for (int x = 0; x < w; ++x) {
				int u, v;
				if (format == 12) { const uint8_t* p = cplane + (size_t)(y / 2) * w + (x / 2) * 2; u = p[0]; v = p[1]; }
				else { const size_t ci = (size_t)(y / 2) * (w / 2) + x / 2; v = cplane[ci]; u = cplane[csize + ci]; }
				const int uu = u - 128, vv = v - 128;
				const int ruv = (1 << (SHIFT - 1)) + CVR * vv, guv = (1 << (SHIFT - 1)) + CVG * vv + CUG * uu, buv = (1 << (SHIFT - 1)) + CUB * uu;
				const int yy = img[(size_t)y * w + x] - 16, yv = (yy < 0 ? 0 : yy) * CY;
				uint8_t* o = rgb + ((size_t)y * w + x) * 3;
				o[0] = sat_u8((yv + ruv) >> SHIFT); o[1] = sat_u8((yv + guv) >> SHIFT); o[2] = sat_u8((yv + buv) >> SHIFT);
			}
		return 0;
	}
	if (format == 4) { for (size_t i = 0; i < n; ++i) { rgb[3 * i] = img[4 * i]; rgb[3 * i + 1] = img[4 * i + 1]; rgb[3 * i + 2] = img[4 * i + 2]; } return 0; }
	memcpy(rgb, img, n * 3);
	return 0;
}

/* cimbard_scan_extract_decode's extract half for a capture in `format` (cimbar_recv_js.cpp:160-179): get_rgb, then Extractor::extract */
int co_extract_fmt(const uint8_t* img, int w, int h, int format, uint8_t* out1024, float* corners8)
{
	uint8_t* rgb = (uint8_t*)malloc((size_t)w * h * 3 + 1);
	int rc = 0;
	if (co_capture_to_rgb(img, w, h, format, rgb) == 0) rc = co_extract(rgb, w, h, out1024, corners8);
	free(rgb);
	return rc;
}
