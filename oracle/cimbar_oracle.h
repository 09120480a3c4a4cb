/* oracle/cimbar_oracle.h -- CPU restatement of libcimbar's mode-B frame decode path.
 *
 * TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (libcimbar_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED against the reference's own code for this path, compiled from /root/reference by
 * oracle/Makefile into oracle/_ref/libcimbar_ref.so (tests/test_oracle_vs_ref.py), and against the reference's
 * known-answer vectors (tests/test_oracle_golden.py). The OpenCV primitives the reference calls (gray conversion,
 * box-mean threshold, filter2D, SVD pseudo-inverse) are NOT in /root/reference; both this file and the cv-shim the
 * reference is compiled against restate OpenCV 4.5.x's published arithmetic. The SVD / pseudo-inverse / von Kries
 * part of that is pinned to the matrices a real OpenCV printed in color_correctionTest.cpp:14-82; the per-pixel part
 * (gray, box mean, filter2D, cv::mean) has no sample-free known-answer test in the reference, so parity AT THAT PART OF
 * THE OPENCV BOUNDARY IS UNPINNED (see DESIGN.md "Oracle").
 */
#ifndef CIMBAR_ORACLE_H
#define CIMBAR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One geometry per build of the oracle (cc -DCO_MODE=67 -> libcimbar_oracle_m67.so), Config.h:19-44 + GridConf.h:121-186:
 * 68 = Conf8x8 ("B", the default), 67 = Conf8x8_mini ("Bm"), 66 = Conf8x8_micro ("Bu"), 4 / 8 = the legacy 4- and 8-colour modes ("4C" / "8C": Conf8x8
 * with symbol and colour bits coupled in one Reed-Solomon stream, Decoder.h:121-161). */
#ifndef CO_MODE
#define CO_MODE 68
#endif
#if CO_MODE == 68
#define CO_IMG_W 1024
#define CO_IMG_H 1024
#define CO_OFFSET 8
#define CO_DIM_X 112
#define CO_DIM_Y 112
#define CO_RS_BLOCK 155
#define CO_RS_PARITY 30
#define CO_CHUNKS_PER_FRAME 12
#elif CO_MODE == 67
#define CO_IMG_W 1024
#define CO_IMG_H 720
#define CO_OFFSET 9
#define CO_DIM_X 112
#define CO_DIM_Y 78
#define CO_RS_BLOCK 179
#define CO_RS_PARITY 36
#define CO_CHUNKS_PER_FRAME 12
#elif CO_MODE == 66                                                /* Conf8x8_micro ("Bu"), GridConf.h:144-166: fountain_chunks_scalar 1 */
#define CO_IMG_W 736
#define CO_IMG_H 637
#define CO_OFFSET 9
#define CO_DIM_X 80
#define CO_DIM_Y 69
#define CO_RS_BLOCK 168
#define CO_RS_PARITY 33
#define CO_CHUNKS_PER_FRAME 6
#elif CO_MODE == 4                                                 /* Config.h:24-29: Conf8x8, 2 colour bits, legacy_mode (coupled decode, old palette), */
#define CO_IMG_W 1024                                              /* fountain_chunks_scalar -10 -> 10 chunks of 750 bytes per frame */
#define CO_IMG_H 1024
#define CO_OFFSET 8
#define CO_DIM_X 112
#define CO_DIM_Y 112
#define CO_RS_BLOCK 155
#define CO_RS_PARITY 30
#define CO_CHUNKS_PER_FRAME 10
#define CO_LEGACY 1
#elif CO_MODE == 8                                                 /* Config.h:30-35: the legacy 8-colour mode: 3 colour bits, 7-bit cells, 70 blocks, */
#define CO_IMG_W 1024                                              /* ten chunks of 875 bytes */
#define CO_IMG_H 1024
#define CO_OFFSET 8
#define CO_DIM_X 112
#define CO_DIM_Y 112
#define CO_RS_BLOCK 155
#define CO_RS_PARITY 30
#define CO_CHUNKS_PER_FRAME 10
#define CO_LEGACY 1
#define CO_COLOR_BITS 3
#else
#error "CO_MODE must be 68, 67, 66, 4 or 8"
#endif
#ifndef CO_LEGACY
#define CO_LEGACY 0
#endif
#ifndef CO_COLOR_BITS
#define CO_COLOR_BITS 2
#endif
#define CO_CELL_BITS (4 + CO_COLOR_BITS)
#define CO_CELLS (CO_DIM_X * CO_DIM_Y - 4 * 6 * 6)               /* 12400 | 8592 */
#define CO_RS_DATA (CO_RS_BLOCK - CO_RS_PARITY)                  /* 125 | 143 | 135 */
#define CO_CHUNK (CO_CELLS * CO_CELL_BITS / 8 / CO_RS_BLOCK * CO_RS_DATA / CO_CHUNKS_PER_FRAME)   /* 625 | 429 | 540 | 750 | 875 */

/* the constants above, for the tests: {mode, image w, image h, cells, chunk bytes, RS block, RS parity, cells per row, cell rows, cell offset} */
void co_geometry(int32_t out10[10]);

/* models the reference's `static thread_local color_correction` (CimbDecoder.cpp:69-73) */
typedef struct co_ccm { float m[9]; int active; } co_ccm;

void co_tile_hashes(uint64_t out16[16]);
void co_cell_positions(int32_t* xy /* 2*CO_CELLS */);
void co_interleave_reverse(uint32_t* out /* CO_CELLS */);
void co_adjacent(int index, int32_t out4[4]);

/* CimbReader.cpp:30-46 + bitmatrix.h:14-46: RGB8 -> packed bitplane (w*h/8 bytes, MSB = leftmost pixel) */
void co_threshold_bitplane(const uint8_t* rgb, int w, int h, int preprocess, uint8_t* bitplane);

/* CimbReader.cpp:139-162 + FloodDecodePositions.cpp + CimbDecoder.cpp:101-147: flood-ordered symbol pass.
 * visit[4*k+0..3] = cell index, drifted x, drifted y, symbol of the k-th decoded cell. dist (optional): error distance.
 * Returns the number of cells visited. */
int co_symbol_pass(const uint8_t* bitplane, int32_t* visit, uint8_t* dist);
int co_last_heap_peak(void);   /* of the last co_symbol_pass on this thread: most live queue entries / pops */
int co_last_heap_pops(void);

/* libcorrect decode.c:299-379 (correct_reed_solomon_decode), GF(2^8) poly 0x187, fcr 1, gap 1 */
int co_rs_decode(const uint8_t* enc, unsigned enc_len, unsigned parity, uint8_t* msg);
/* libcorrect encode.c:3-34 */
int co_rs_encode(const uint8_t* msg, unsigned msg_len, unsigned parity, uint8_t* enc);

/* CimbDecoder.cpp:168-200 */
unsigned co_best_color(float r, float g, float b, const co_ccm* ccm);

/* chromatic_adaptation/color_correction.h:26-39, :11-24, :64-68 -- exported for the reference's known-answer tests */
void co_moore_penrose_lsm(const float* actual, const float* desired, int rows, float out9[9]);
void co_von_kries_ccm(const float white[3], float out9[9]);
void co_ccm_transform(const float m[9], float r, float g, float b, float out3[3]);

/* Decoder::decode_fountain (Decoder.h:171-189) for one 1024x1024 RGB8 frame.
 * chunks: 12*625 bytes, slot j = fountain chunk j (zero-filled if dropped); *good_mask bit j = chunk j delivered.
 * ccm: in/out carried colour-correction state. Returns good bytes (625 * popcount(mask)). */
int co_decode_fountain(const uint8_t* rgb, int w, int h, int preprocess, int color_correction, co_ccm* ccm,
                       uint8_t* chunks, uint32_t* good_mask);

/* Decoder::decode (Decoder.h:163-169) for one frame into a plain stream: bytes = 60 x 125 RS outputs, a failed block as zeros
 * (reed_solomon_stream.h:62-74,96-107); block_ok (60 bytes, may be NULL) = 1 where libcorrect succeeded. Returns 7500. */
int co_decode_plain(const uint8_t* rgb, int w, int h, int preprocess, int color_correction, co_ccm* ccm,
                    uint8_t* bytes, uint8_t* block_ok);

/* stage outputs of the last co_decode_fountain call on this thread (for stage-level parity tests) */
const uint8_t* co_last_symbols(void);   /* CO_CELLS bytes, by cell index */
const uint8_t* co_last_colors(void);    /* CO_CELLS bytes, by cell index */
const int32_t* co_last_positions(void); /* 2*CO_CELLS, drifted x,y by cell index */

/* ---- the stage in front of the decoder (oracle/cimbar_oracle_extract.c). Parity unpinned at the OpenCV boundary. */
/* Scanner::preprocess_image(img, fast=true), Scanner.h:148-165: gray -> 3x3|5x5 Gaussian -> Otsu. out: w*h bytes 0/255. Returns the threshold. */
int co_gray_blur(const uint8_t* rgb, int w, int h, uint8_t* out, int* hist_out);
int co_scan_preprocess(const uint8_t* rgb, int w, int h, uint8_t* out);
/* cv::getPerspectiveTransform as called by Deskewer::deskew, Deskewer.h:36. Row-major 3x3 into m9. */
int co_perspective_transform(const float* src8, const float* dst8, double* m9);
void co_deskew_points(float* dst8);
/* cv::warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0), Deskewer.h:38: RGB8 sw x sh -> RGB8 width x height */
int co_warp_perspective(const uint8_t* rgb, int sw, int sh, const double* m9, uint8_t* out, int width, int height);
/* Deskewer::deskew for mode B from the four corners (top-left, top-right, bottom-left, bottom-right; x, y) */
int co_deskew(const uint8_t* rgb, int sw, int sh, const float* corners8, uint8_t* out1024);

/* Scanner::scan (Scanner.cpp:183-202) on the 0/255 image: up to 4 anchors {x, xmax, y, ymax} (top-left, top-right, bottom-left, bottom-right); returns the count */
int co_scan_anchors(const uint8_t* binary, int w, int h, int32_t* anchors16);
/* Extractor::extract (Extractor.h:29-45): 0 failure / 1 success / 2 needs sharpen; corners8 = Corners::all(); out = deskewed 1024x1024 RGB8 */
int co_extract(const uint8_t* rgb, int w, int h, uint8_t* out1024, float* corners8);

/* get_rgb (cimbar_js/cimbar_recv_js.cpp:94-120): a capture in the C ABI's `format` (3 RGB, 4 RGBA, 12 NV12, 420 = COLOR_YUV420p2RGB) -> RGB8.
 * co_capture_bytes: the bytes such a capture occupies (0: the format cannot hold that size). co_capture_to_rgb returns 0, or -1 for such a size. */
size_t co_capture_bytes(int w, int h, int format);
int co_capture_to_rgb(const uint8_t* img, int w, int h, int format, uint8_t* rgb);
int co_extract_fmt(const uint8_t* img, int w, int h, int format, uint8_t* out1024, float* corners8);

#ifdef __cplusplus
}
#endif
#endif
