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
int co_scan_preprocess(const uint8_t* rgb, int w, int h, uint8_t* out)
{
	const size_t n = (size_t)w * h;
	unsigned unit = (unsigned)(w < h ? w : h);
	unit = next_pow2_plus_one((unsigned)(unit * 0.002));
	if (unit != 3 && unit != 5) return -1;
	const int r = (int)unit / 2, shift = r == 1 ? 4 : 8;
	static const int k3[3] = {1, 2, 1}, k5[5] = {1, 4, 6, 4, 1};
	const int* k = r == 1 ? k3 : k5;

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
	const float size = 1024, anchor = 30;   /* Config::image_size_x/y(), Config::anchor_size() for mode B; padding 0 */
	dst8[0] = anchor; dst8[1] = anchor;
	dst8[2] = size - anchor; dst8[3] = anchor;
	dst8[4] = anchor; dst8[5] = size - anchor;
	dst8[6] = size - anchor; dst8[7] = size - anchor;
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

/* Deskewer::deskew for mode B (1024x1024 output, anchor 30, padding 0) from the corners Corners::all() would return */
int co_deskew(const uint8_t* rgb, int sw, int sh, const float* corners8, uint8_t* out1024)
{
	float dst8[8];
	double m9[9];
	co_deskew_points(dst8);
	co_perspective_transform(corners8, dst8, m9);
	return co_warp_perspective(rgb, sw, sh, m9, out1024, 1024, 1024);
}
