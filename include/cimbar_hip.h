/* cimbar_hip.h -- C ABI of the MI355X (gfx950) cimbar mode-B frame-decode path.
 *
 * Drop-in boundary: everything between "a deskewed 1024x1024 RGB8 frame" and "the <=12 fountain chunks of 625 bytes
 * that libcimbar hands to fountain_decoder_sink::write". It replaces, for that path only, what the reference does in
 *     Decoder::decode_fountain            /root/reference/src/lib/encoder/Decoder.h:171-189 (-> do_decode :60-118)
 *     CimbReader / CimbDecoder            src/lib/cimb_translator/CimbReader.cpp:107-280, CimbDecoder.cpp:101-217
 *     reed_solomon_stream / aligned_stream src/lib/encoder/reed_solomon_stream.h:54-77, aligned_stream.h:39-119
 * and mirrors the shape of the reference's own C ABI for the same step,
 *     cimbard_get_bufsize / cimbard_scan_extract_decode / cimbard_configure_decode
 *                                          src/lib/cimbar_js/cimbar_recv_js.h:16-17,36; cimbar_recv_js.cpp:143-189
 * (minus the Scanner/Extractor call, which is upstream of this path).
 *
 * Conventions: plain C types, caller-allocated buffers, no exceptions across the boundary. Functions returning int
 * return >= 0 on success and a negative CIMBAR_HIP_E* code on failure. One context serves one host thread at a time
 * and carries the colour-correction matrix from frame to frame exactly like the reference's `static thread_local`
 * CCM (CimbDecoder.cpp:69-73). There is NO CPU fallback: if no gfx950 device/kernel image is usable, create() fails.
 */
#ifndef CIMBAR_HIP_H
#define CIMBAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIMBAR_HIP_FRAME_DIM 1024          /* Conf8x8 image_size_x/y, GridConf.h:130-131 */
#define CIMBAR_HIP_CELLS 12400             /* GridConf.h:42-45 */
#define CIMBAR_HIP_CHUNK_SIZE 625          /* Config::fountain_chunk_size(), GridConf.h:63-71 */
#define CIMBAR_HIP_CHUNKS_PER_FRAME 12     /* Config::fountain_chunks_per_frame(6), GridConf.h:54-61 */
#define CIMBAR_HIP_FRAME_BYTES (CIMBAR_HIP_CHUNK_SIZE * CIMBAR_HIP_CHUNKS_PER_FRAME) /* 7500 = cimbard_get_bufsize() in mode B -- MODE B ONLY */
/* The largest chunk space any mode's frame needs: mode 8 (the legacy 8-colour mode) delivers 10 chunks of 875 bytes = 8750 (modes 68 / 4:
 * 7500, 67: 5148, 66: 3240). Size fixed buffers with THIS, or ask cimbar_hip_ctx_bufsize(ctx) -- never with CIMBAR_HIP_FRAME_BYTES unless the
 * context is known to be mode B's. */
#define CIMBAR_HIP_MAX_FRAME_BYTES 8750

enum {
	CIMBAR_HIP_OK = 0,
	CIMBAR_HIP_EINVAL = -1,       /* bad argument (null pointer, n <= 0, unsupported mode) */
	CIMBAR_HIP_EDIM = -2,         /* frame is not 1024x1024 RGB8 (cf. CimbReader::_good, CimbReader.cpp:119) */
	CIMBAR_HIP_ENODEVICE = -3,    /* no usable gfx950 device / kernel image */
	CIMBAR_HIP_EHIP = -4,         /* a HIP runtime call failed; see cimbar_hip_last_error() */
	CIMBAR_HIP_ENOMEM = -5
};

/* where a buffer argument lives */
enum { CIMBAR_HIP_MEM_HOST = 0, CIMBAR_HIP_MEM_DEVICE = 1 };

typedef struct cimbar_hip_ctx cimbar_hip_ctx;

/* cimbard_configure_decode(mode) + `Decoder dec;` (cimbar_recv_js.cpp, Config::update, Config.h:19-50). Modes built: 68 ("B", Conf8x8,
 * 1024x1024, GridConf.h:121-142; 0 selects it too, Config::temp_conf's default), 67 ("Bm", Conf8x8_mini, 1024x720, GridConf.h:168-189) and
 * 66 ("Bu", Conf8x8_micro, 736x637, GridConf.h:144-166), and 4 (the legacy 4-colour mode, Config.h:24-29: mode B's grid with the coupled decode
 * of Decoder.h:121-161 -- one Reed-Solomon stream of 6-bit cells, the old palette, no header-derived colour correction) and 8 (the legacy
 * 8-colour mode, Config.h:30-35: 7-bit cells, 70 blocks, 10 * 875 bytes) -- every mode Config::temp_conf knows. Any other value selects mode
 * 68, exactly like Config::temp_conf's `default:` branch (Config.h:41-43): cimbard_configure_decode(5) upstream gives a working mode-B
 * decoder, and so does cimbar_hip_create(dev, 5) (cimbar_hip_geometry then reports mode 68). `device` is a HIP device ordinal. Everywhere below "frame" means an
 * image_size_x x image_size_y RGB8 image of the context's mode, "12 * 625" the mode's chunks-per-frame * chunk size (12 * 429 in mode 67,
 * 6 * 540 in mode 66, 10 * 750 in mode 4; the mask has as many bits) and "60 blocks of 125" its RS layout (36 of 143; 24 of 135):
 * cimbar_hip_geometry reports the numbers. */
int cimbar_hip_create(int device, int mode_val, cimbar_hip_ctx** out);
void cimbar_hip_destroy(cimbar_hip_ctx* ctx);

/* cimbard_get_bufsize() (cimbar_recv_js.cpp:143-146) = fountain_chunks_per_frame() * fountain_chunk_size() of the ACTIVE configuration. The
 * reference's configuration is a thread_local; here it lives in the context, so the faithful counterpart takes one:
 *   cimbar_hip_ctx_bufsize(ctx) : the chunk space one frame of THIS context needs (7500 / 5148 / 3240 / 7500 / 8750 for modes 68 / 67 / 66 / 4 / 8)
 *   cimbar_hip_bufsize()        : the same for the default configuration (mode B, what a thread that never called Config::update has): 7500
 *   cimbar_hip_mode_bufsize(m)  : the same for the configuration cimbard_configure_decode(m) selects, without a context or a device (what a
 *                                 cimbard_get_bufsize() replacement answers before the first frame arrives: libcimbar_recv_hip.so) */
int cimbar_hip_ctx_bufsize(const cimbar_hip_ctx* ctx);
int cimbar_hip_bufsize(void);
int cimbar_hip_mode_bufsize(int mode_val);

/* The grid a context was created for -- the Config:: getters the reference's callers size their buffers with (Config.h:52-165):
 * out = {mode, image_size_x, image_size_y, total_cells, fountain_chunks_per_frame, fountain_chunk_size, RS blocks per frame (symbol + colour),
 *        ecc_block_size, ecc_bytes, cells_per_col_x, cells_per_col_y, cell_offset}. Returns CIMBAR_HIP_GEOMETRY_WORDS. */
enum { CIMBAR_HIP_GEOMETRY_WORDS = 12 };
int cimbar_hip_geometry(const cimbar_hip_ctx* ctx, int32_t out[CIMBAR_HIP_GEOMETRY_WORDS]);

/* The 16 symbol-tile hashes as cimbar_hip_create computes them from the embedded tile bitmaps -- what CimbDecoder's constructor does
 * (CimbDecoder.cpp:58-66,87-99: getTile -> average_hash). Host-only arithmetic (works without a device); returns 16. */
int cimbar_hip_tile_hashes(uint64_t out16[16]);

/* the HIP device ordinal the context lives on */
int cimbar_hip_device(const cimbar_hip_ctx* ctx);

/* human-readable text of the last failure on this context (never NULL) */
const char* cimbar_hip_last_error(const cimbar_hip_ctx* ctx);

/* Decoder::decode_fountain(img, sink, should_preprocess, color_correction) for ONE host-resident frame.
 *   rgb        : height rows of `stride` bytes, width*3 used (RGB8, as cv::Mat CV_8UC3 after BGR2RGB, cimbar.cpp:132-133)
 *   chunks     : cimbar_hip_ctx_bufsize(ctx) bytes (12*625 in mode B; at most CIMBAR_HIP_MAX_FRAME_BYTES); slot j holds fountain chunk j of the
 *                frame, zero-filled if the chunk was dropped
 *   good_mask  : bit j set <=> aligned_stream delivered chunk j to the sink (aligned_stream.h:62-85)
 * Returns the reference's return value: cumulative good bytes = 625 * popcount(mask) (Decoder.h:116-117).
 * Image size, as CimbReader's constructor treats it (CimbReader.cpp:107-126):
 *   width x height == image_size_x x image_size_y : the ordinary case (and the only one the batch entry points take);
 *   larger in either direction : the grid sits _gridPadding = min(width - image_size_x, height - image_size_y) / 2 pixels in, in x and in y, and
 *                                the threshold pass sees the real pixels around it (a single-image path, not tuned for throughput);
 *   smaller : the reference reads no cell at all, Reed-Solomon then "decodes" its all-zero buffers and every chunk is delivered as zeros --
 *             the call returns the full byte count, mask 0xFFF and zero-filled chunks, exactly like the reference (a fountain sink drops them). */
int cimbar_hip_decode_frame(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, size_t stride,
                            int should_preprocess, int color_correction, uint8_t* chunks, uint32_t* good_mask);

/* The same call with frames in flight -- the shape of the reference's decode loop (cimbar.cpp:124-171: one Decoder::decode_fountain per image,
 * the next image being read while this one decodes) and of web/recv-worker.js's frames-keep-arriving loop: frame k+1's host-to-device copy
 * runs beside frame k's kernels, so one context sustains the PCIe copy rate instead of copy + kernels + copy-back per frame.
 *   cimbar_hip_decode_frame_async : starts the frame and returns a ticket >= 0 (negative: an error code, nothing started). At most
 *                     cimbar_hip_pipeline_depth(ctx) frames are in flight; starting one more first completes the oldest (its chunks and mask land
 *                     in the buffers it was started with; its return value stays available to _wait for the next 16 tickets).
 *                     `rgb` in page-locked memory (hipHostMalloc / hipHostRegister) is copied from where it lies and must stay untouched until the
 *                     frame's _wait; pageable memory has been consumed when the call returns (a dense image through the runtime's own pageable copy, a
 *                     strided one -- a cv::Mat ROI -- through the context's page-locked staging) and may be reused or freed at once. `chunks` / `good_mask` are written by _wait (or by the completion described above), never before.
 *                     Images of another size than the frame (CimbReader.cpp:107-126's padded / too-small cases) are decoded synchronously behind
 *                     everything in flight and still get a ticket.
 *   cimbar_hip_decode_frame_wait  : blocks until that frame is complete and returns what cimbar_hip_decode_frame would have returned.
 * Frames are decoded in ticket order (the colour-correction matrix carries over from frame to frame exactly as in the synchronous call), so
 * waiting in ticket order hands the chunks to a sink in the order the reference's loop would. cimbar_hip_decode_frame IS _async + _wait.
 * Mixing with the batch entry points on one context: the frame slots ride on the pipeline's streams and scratch sets and are ordered among
 * themselves and behind cimbar_hip_decode_batch_pipelined; a batch call with DEVICE outputs on a stream of the caller's only enqueues, shares the
 * carried matrix and the flood scratch with the frame slots and is NOT ordered against them -- synchronise that stream before the next
 * cimbar_hip_decode_frame_async (a batch call with host outputs has synchronised when it returns). */
long long cimbar_hip_decode_frame_async(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, size_t stride,
                                        int should_preprocess, int color_correction, uint8_t* chunks, uint32_t* good_mask);
int cimbar_hip_decode_frame_wait(cimbar_hip_ctx* ctx, long long ticket);

/* The same for `n` independent frames, decoded in frame order (frame f's CCM carry-over sees frames < f).
 *   rgb        : n densely packed 1024*1024*3 frames, in host or device memory (rgb_mem)
 *   chunks     : n*7500 bytes, masks: n words, in host or device memory (out_mem)
 *   hip_stream : a hipStream_t to enqueue on. NULL = the null stream when a device buffer is involved (ordinary HIP
 *                semantics: ordered after the work that produced the frames there), the context's own stream when
 *                everything is host memory. With device outputs the call only enqueues work and returns 0; the caller
 *                synchronises the stream. With host outputs it synchronises and returns the total good bytes over the batch.
 */
int64_t cimbar_hip_decode_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, int n, int rgb_mem, int should_preprocess,
                                int color_correction, uint8_t* chunks, uint32_t* masks, int out_mem, void* hip_stream);

/* A continuous stream of batches (device buffers only): the same work and the same results as cimbar_hip_decode_batch, but the call only
 * records "the frames are ready" on `hip_stream` and runs the batch on one of D = cimbar_hip_pipeline_depth() streams the context owns,
 * consecutive batches on consecutive streams. Up to D batches are then in flight at once and the threshold pass of one batch -- the
 * HBM-bound 70 % of the work -- overlaps the short, latency-bound kernels of the others; the colour-correction matrix still carries
 * over from batch to batch in order (a batch's colour pass waits for the end of the batch before it). The reference's receive loop has
 * the same shape: frames keep arriving while earlier ones are decoded (cimbar_recv_js.cpp:143-189 under web/recv.js's worker pool).
 * Contract: at most D batches in flight; the rgb / chunks / masks buffers of a batch must stay untouched until a
 * cimbar_hip_pipeline_wait that covers it has been enqueued and reached. Any other entry point of the context waits for the pipeline.
 *   cimbar_hip_pipeline_wait(ctx, stream, keep_newest): `stream` waits for every pipelined batch issued so far except the `keep_newest`
 *   most recent ones (0 = all of them; D-1 = only the oldest that can still be in flight: consume batch k-D+1 right after issuing
 *   batch k and the pipeline stays full). Enqueue-only, returns 0. */
int cimbar_hip_decode_batch_pipelined(cimbar_hip_ctx* ctx, const uint8_t* rgb, int n, int should_preprocess, int color_correction,
                                      uint8_t* chunks, uint32_t* masks, void* hip_stream);
int cimbar_hip_pipeline_wait(cimbar_hip_ctx* ctx, void* hip_stream, int keep_newest);
int cimbar_hip_pipeline_depth(const cimbar_hip_ctx* ctx);

/* Decoder::decode(img, ostream, should_preprocess, color_correction) (src/lib/encoder/Decoder.h:163-169) -- the `./cimbar --no-fountain`
 * path (cimbar.cpp:270-272) -- for n frames: no aligned_stream, every 125-byte Reed-Solomon output is written where it falls and a
 * block libcorrect could not decode is written as 125 zero bytes (reed_solomon_stream.h:62-74,96-107).
 *   bytes    : n * 7500 bytes (60 blocks of 125 per frame: 40 from the symbol bits, then 20 from the colour bits)
 *   block_ok : n * 60 bytes, 1 = the block decoded; may be NULL. Same memory kind as `bytes`.
 * No fountain header reaches the reader on this path, so color_correction == 2 keeps whatever matrix the context carries
 * (CimbReader.cpp:169-180). Host outputs: synchronises and returns n * 7500 (what the stream's tellp() advanced by, Decoder.h:116-117);
 * device outputs: enqueues and returns 0. Buffers, stream and errors as for cimbar_hip_decode_batch. */
int64_t cimbar_hip_decode_plain_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, int n, int rgb_mem, int should_preprocess,
                                      int color_correction, uint8_t* bytes, uint8_t* block_ok, int out_mem, void* hip_stream);

/* Clears the carried colour-correction state (what a fresh thread starts with in the reference). */
int cimbar_hip_reset_ccm(cimbar_hip_ctx* ctx);
/* Current carried CCM, row-major 3x3; returns 1 if active, 0 if not (CimbDecoder::get_ccm, CimbDecoder.cpp:76-80). */
int cimbar_hip_get_ccm(cimbar_hip_ctx* ctx, float out9[9]);
/* CimbDecoder::update_color_correction (CimbDecoder.cpp:82-85; what DecoderPlus::load_ccm feeds from `--color-correction-file`,
 * DecoderPlus.h:32-45, cimbar.cpp:265-266): the carried matrix becomes m9 (row-major 3x3) and is active from the next frame on, until a frame
 * derives its own (color_correction 2 with a decoded header) or cimbar_hip_reset_ccm. Waits for batches in flight. */
int cimbar_hip_set_ccm(cimbar_hip_ctx* ctx, const float m9[9]);

/* ---- the encode half ("next" row of the scope table: on-device frame synthesiser) --------------------------------------------
 * Encoder::encode_next (src/lib/encoder/Encoder.h:69-129) for n frames at once: each frame takes 7500 payload bytes (the 60
 * reads of 125 bytes a fountain_encoder_stream / ifstream would have served), RS(155,125)-encodes them (libcorrect encode.c:3-34),
 * stripes the 4 symbol bits and 2 colour bits of every cell through the interleave (CimbWriter.cpp:84-95, Interleave.h:8-24) and
 * pastes tile colour*16+symbol (Common.cpp:150-171) at every cell of a 1024x1024 RGB8 frame.
 * The background / anchors / guides come from a template frame the caller supplies once (an empty CimbWriter image,
 * CimbWriter.cpp:39-77): this library ships no bitmap assets of its own. */
int cimbar_hip_set_template(cimbar_hip_ctx* ctx, const uint8_t* rgb_template, int mem);
int cimbar_hip_encode_batch(cimbar_hip_ctx* ctx, const uint8_t* payload, int n, int payload_mem, uint8_t* rgb_out, int rgb_mem,
                            void* hip_stream);

/* ---- the stage in front of the decoder ("next" row 2 of the scope table: Scanner's image preparation + Deskewer) ----------------------
 * The reference turns a camera capture into the decoder's 1024x1024 frame with Extractor::extract (src/lib/extractor/Extractor.h:29-45):
 * Scanner (gray -> small Gaussian -> Otsu threshold, Scanner.h:148-165; then a sparse scan-line search for the four anchors on that
 * binary image, Scanner.h:277-405) and Deskewer::deskew (cv::getPerspectiveTransform + cv::warpPerspective INTER_LINEAR,
 * Deskewer.h:26-40). These two calls are the image passes on their own (a caller with its own anchor search, or the adapter's
 * cimbar_amd::Deskewer); the whole of Extractor::extract, anchor search included, is cimbar_hip_extract_batch below. The warp's output can
 * stay in device memory for cimbar_hip_decode_batch. Any width x height RGB8 capture (densely packed frames); OpenCV's arithmetic is restated,
 * see DESIGN.md.
 *   cimbar_hip_scan_preprocess : n captures -> n * width * height bytes (0 / 255) = Scanner::preprocess_image(img, fast = true);
 *                                thresholds (n ints, may be NULL) receives the Otsu thresholds. Blur unit as in Scanner.h:157-159 (3x3 below 1500 px on the short side, 5x5 below 2500, 9x9 below 4500, 17x17 below 8500); 8500 px and more: EDIM.
 *   cimbar_hip_deskew_batch    : corners = n * 8 floats in HOST memory, per capture top-left, top-right, bottom-left, bottom-right (x, y)
 *                                exactly as Corners::all() returns them (Corners.h:45-53); frames = n * 1024*1024*3 bytes = what
 *                                Deskewer(0, {1024,1024}, 30).deskew(img, corners) returns.
 * Buffers, stream and errors as for cimbar_hip_decode_batch (host outputs: synchronises; device outputs: enqueues and returns 0). */
int cimbar_hip_scan_preprocess(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, int n, int rgb_mem,
                               uint8_t* binary, int* thresholds, int out_mem, void* hip_stream);
int cimbar_hip_deskew_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, int n, int rgb_mem,
                            const float* corners, uint8_t* frames, int out_mem, void* hip_stream);

/* Extractor::extract (src/lib/extractor/Extractor.h:29-45) for n captures, entirely on the device: Scanner (image preparation AND the
 * anchor search: Scanner.h:277-405, Scanner.cpp:52-202, ScanState.h:21-104) -> Corners (Corners.h:45-73) -> Deskewer::deskew.
 *   status  : n ints, Extractor::FAILURE 0 / SUCCESS 1 / NEEDS_SHARPEN 2 (Extractor.h:18-20). A capture whose anchor search overflows the fast
 *             kernels' fixed-size work lists (hundreds of anchor-like patterns on one scan line) is searched again serially with lists of
 *             16 384 entries (up to 16 such captures per batch); -1 = those overflowed too, treated as a failure
 *   corners : n * 8 floats, Corners::all() (top-left, top-right, bottom-left, bottom-right; x, y); may be NULL. Unset where status <= 0
 *   frames  : n * 1024*1024*3 bytes, black where status <= 0
 * status / corners / frames share out_mem. Buffers, stream and errors as for cimbar_hip_decode_batch. */
int cimbar_hip_extract_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, int n, int rgb_mem, uint8_t* frames,
                             int* status, float* corners, int out_mem, void* hip_stream);

/* cimbard_scan_extract_decode (src/lib/cimbar_js/cimbar_recv_js.cpp:148-189) / the body of cimbar.cpp's decode loop (:124-162) for n
 * captures: extract, then Decoder::decode_fountain on what came out, the deskewed frames never leaving the device.
 *   preprocess : 1 sharpen every frame, 0 none, anything else ("-1 == guess", cimbar.cpp:190) where the extractor said NEEDS_SHARPEN
 *   chunks / masks : as for cimbar_hip_decode_batch; a capture whose extraction failed delivers nothing (mask 0, slots zeroed)
 *   status     : as for cimbar_hip_extract_batch; may be NULL
 * Host outputs: synchronises and returns the total good bytes; device outputs: enqueues and returns 0. */
int64_t cimbar_hip_scan_extract_decode_batch(cimbar_hip_ctx* ctx, const uint8_t* rgb, unsigned width, unsigned height, int n, int rgb_mem,
                                             int preprocess, int color_correction, uint8_t* chunks, uint32_t* masks, int* status, int out_mem,
                                             void* hip_stream);

/* ---- the capture's pixel format: the `format` argument of the reference's own C ABI for this step ----------------------------------------
 *     int cimbard_scan_extract_decode(const unsigned char* imgdata, unsigned imgw, unsigned imgh, int format, unsigned char* bufspace, unsigned bufsize)
 *                                          src/lib/cimbar_js/cimbar_recv_js.h:17; get_rgb, cimbar_recv_js.cpp:94-120; `format <= 0` is 3, :150-151
 * whose only camera caller hands over VideoFrames as they come: NV12 (12), I420 (420) or RGBA (4) (web/recv-worker.js:38-47, web/recv.js:110,362-371).
 * The four entry points above with that argument in the reference's position (after the height). `img` = n captures of
 * cimbar_hip_capture_bytes(width, height, format) bytes each, back to back:
 *   3 (and <= 0, and any value the reference's `default:` lets through)  RGB8, width * height * 3
 *   4    RGBA8, width * height * 4; the alpha byte is dropped (cv::COLOR_RGBA2RGB)
 *   12   NV12: width * height luma bytes, then height / 2 rows of width bytes (U, V, U, V ...)        (cv::COLOR_YUV2RGB_NV12)
 *   420  three planes: width * height luma, then two (width / 2) x (height / 2) chroma planes, read the way the reference's conversion code
 *        cv::COLOR_YUV420p2RGB reads them -- OpenCV defines it as COLOR_YUV2RGB_YV12: the FIRST chroma plane is V. (An I420 VideoFrame has U
 *        first; upstream decodes such captures with red and blue exchanged and leaves it to the header-derived colour correction. Kept.)
 * YUV -> RGB is OpenCV's fixed-point BT.601, gray / blur / warp then see exactly the RGB image get_rgb would have built -- but no
 * such image is ever written: the scan and warp kernels convert as they load, so a 1080p NV12 capture costs 3.1 MB of PCIe and HBM reads
 * instead of 6.2. 12 and 420 need an even width and height (cv::cvtColor asserts it; upstream throws): CIMBAR_HIP_EDIM, and
 * cimbar_hip_capture_bytes returns 0. Everything else as for the entry point without the suffix (which is format 3).
 * cimbard_scan_extract_decode(img, w, h, format, buf, size) is then
 *   cimbar_hip_scan_extract_decode_batch_fmt(ctx, img, w, h, format, 1, CIMBAR_HIP_MEM_HOST, 1, 2, chunks, &mask, &status, CIMBAR_HIP_MEM_HOST, NULL)
 * (upstream always sharpens on this path: `bool shouldPreprocess = true`, cimbar_recv_js.cpp:166) with status 0 -> return -3 and the chunks
 * whose mask bit is set packed front to back (escrow_buffer_writer); INTEGRATION.md has the wrapper. */
size_t cimbar_hip_capture_bytes(unsigned width, unsigned height, int format);
int cimbar_hip_scan_preprocess_fmt(cimbar_hip_ctx* ctx, const uint8_t* img, unsigned width, unsigned height, int format, int n, int img_mem,
                                   uint8_t* binary, int* thresholds, int out_mem, void* hip_stream);
int cimbar_hip_deskew_batch_fmt(cimbar_hip_ctx* ctx, const uint8_t* img, unsigned width, unsigned height, int format, int n, int img_mem,
                                const float* corners, uint8_t* frames, int out_mem, void* hip_stream);
int cimbar_hip_extract_batch_fmt(cimbar_hip_ctx* ctx, const uint8_t* img, unsigned width, unsigned height, int format, int n, int img_mem,
                                 uint8_t* frames, int* status, float* corners, int out_mem, void* hip_stream);
int64_t cimbar_hip_scan_extract_decode_batch_fmt(cimbar_hip_ctx* ctx, const uint8_t* img, unsigned width, unsigned height, int format, int n,
                                                 int img_mem, int preprocess, int color_correction, uint8_t* chunks, uint32_t* masks, int* status,
                                                 int out_mem, void* hip_stream);

/* ---- multi-GPU: the one exchange step ------------------------------------------------------------------------------------
 * Frames are independent, so each GPU decodes its own slab with its own context; afterwards every rank's n x (7500 chunk bytes + 1 mask
 * word) are gathered on `root` in rank order (== frame order for contiguous slabs), where the host feeds the single fountain_decoder_sink
 * -- the shape of the reference's worker pool -> one sink (web/recv-worker.js:47-64, web/recv.js:36). The gather is ncclGather over RCCL /
 * xGMI (/opt/rocm/include/rccl/rccl.h:745); RCCL is loaded on first use, the library does not link against it.
 *   comm_init_all   : one process driving ndev GPUs (one context + one comm per device; call gather_chunks from one thread per device or
 *                     inside the caller's own group). devices NULL = 0..ndev-1. Fills out[0..ndev).
 *   comm_unique_id / comm_init_rank : one process per GPU; rank 0 makes the 128-byte id and hands it to the others by its own means.
 *   gather_chunks   : chunks / masks: this rank's n frames (device memory); all_chunks / all_masks: nranks * n frames on root (device
 *                     memory, may be NULL elsewhere). Enqueued on hip_stream; returns 0.
 *   comm_info       : the communicator's size and this member's rank as RCCL reports them (ncclCommCount / ncclCommUserRank): what a
 *                     benchmark line quotes so that it says itself how many ranks the exchange ran over. */
typedef struct cimbar_hip_comm cimbar_hip_comm;
int cimbar_hip_comm_init_all(int ndev, const int* devices, cimbar_hip_comm** out);
int cimbar_hip_comm_unique_id(uint8_t id128[128]);
int cimbar_hip_comm_init_rank(const uint8_t id128[128], int nranks, int rank, int device, cimbar_hip_comm** out);
int cimbar_hip_comm_info(cimbar_hip_comm* comm, int* nranks, int* rank);
void cimbar_hip_comm_destroy(cimbar_hip_comm* comm);
int cimbar_hip_gather_chunks(cimbar_hip_ctx* ctx, cimbar_hip_comm* comm, int root, const uint8_t* chunks, const uint32_t* masks, int n,
                             uint8_t* all_chunks, uint32_t* all_masks, void* hip_stream);
/* The same exchange for the batch issued LAST through cimbar_hip_decode_batch_pipelined, enqueued on that batch's own pipeline stream right behind its
 * kernels (chunks / masks = the buffers that call was given): no stream or event of the caller's is involved, and cimbar_hip_pipeline_wait -- which a
 * consumer of the batch calls anyway -- then also covers all_chunks / all_masks on the root. Every rank calls it after every pipelined batch, in the same
 * order. This is what bench.py's N > 1 loop issues per step. CIMBAR_HIP_EINVAL if no pipelined batch has been issued on the context. */
int cimbar_hip_pipeline_gather(cimbar_hip_ctx* ctx, cimbar_hip_comm* comm, int root, const uint8_t* chunks, const uint32_t* masks, int n,
                               uint8_t* all_chunks, uint32_t* all_masks);

/* ---- PNG decode on the device (what cv::imread does in front of the decoder, cimbar.cpp:132-133) --------------------
 * A batch of PNG images whose zlib streams (the concatenated IDAT payloads) already sit in device memory -> dense RGB8 frames in device
 * memory, without the decoded pixels ever crossing PCIe: inflate (one wavefront per image, Huffman decode + LZ77 window in LDS, Adler-32
 * checked) and the scanline un-filter (rows skewed over the lanes). 8 bits per sample, non-interlaced, colour types 0 (gray, replicated),
 * 2 (RGB), 3 (palette), 6 (RGBA, alpha dropped) -- what cv::imread(IMREAD_COLOR) + BGR2RGB gives; at most 2048 pixels wide (frames are
 * 1024 or 736). Context-free: `device` is a HIP ordinal.
 *   d_zbuf / zbuf_bytes : the streams (and palettes), device memory. The kernels read whole dwords: every stream's end rounded up to a multiple
 *                         of 4 must lie inside zbuf_bytes (checked; an image whose padded end does not is refused with EHEADER)
 *   d_desc              : n descriptors, device memory
 *   d_scratch           : n * scratch_stride bytes for the filtered scanlines; scratch_stride >= cimbar_hip_png_scratch_bytes(), multiple of 16
 *   d_rgb               : n * rgb_stride bytes; image i's width*height*3 bytes start at i * rgb_stride
 *   d_status            : n words: 0 or a CIMBAR_HIP_PNG_E* code per image (a refused image leaves its rgb slot undefined)
 * Enqueued on hip_stream; returns 0, or a negative code if the launch itself failed. libcimbar_ingest.so's device mode
 * (include/cimbar_ingest.h) is the host side that parses the files and fills these buffers. */
enum {
	CIMBAR_HIP_PNG_EHEADER = -30,   /* descriptor / zlib header not acceptable (colour type, size, window, preset dictionary, alignment) */
	CIMBAR_HIP_PNG_ESTREAM = -31,   /* invalid deflate stream (block type, stored length, distance too far back, invalid code, truncated, filter type) */
	CIMBAR_HIP_PNG_ECODES = -32,    /* over-subscribed or incomplete Huffman code set (zlib's inflate_table rules) */
	CIMBAR_HIP_PNG_ESIZE = -33,     /* the stream inflates to more or fewer bytes than height * (1 + width * bytes per pixel), or the slot is too small */
	CIMBAR_HIP_PNG_ECHECK = -34     /* Adler-32 mismatch */
};
typedef struct cimbar_hip_png_desc {
	uint64_t zoff;        /* byte offset of the image's zlib stream in d_zbuf, a multiple of 16 */
	uint32_t zlen;        /* its length, below 256 MiB (bit positions are 32-bit) */
	uint32_t width, height;
	uint32_t color_type;  /* 0, 2, 3, 6 */
	uint32_t pal_off;     /* colour type 3: byte offset in d_zbuf of 256 RGB palette entries (768 bytes) */
	uint32_t reserved;
} cimbar_hip_png_desc;
size_t cimbar_hip_png_scratch_bytes(unsigned width, unsigned height, unsigned color_type);
int cimbar_hip_png_decode_batch(int device, const uint8_t* d_zbuf, size_t zbuf_bytes, const cimbar_hip_png_desc* d_desc, int n, uint8_t* d_scratch,
                                size_t scratch_stride, uint8_t* d_rgb, size_t rgb_stride, int32_t* d_status, void* hip_stream);
/* the same with a hint about the launch: 0 = decide by n (what the call above does), 1 = one stream per wavefront (shortest time for a lone
 * launch of up to a few thousand images), 4 = "8192+ images are in flight", whether in one launch or in several concurrent ones (what the
 * ingest library's device mode keeps going). For 4 (and for 0 with n >= 8192) the DEVICE picks the inflate kernel: streams of long matches
 * (frames as Pillow writes them) are fastest four to a wavefront, streams with short literal codes (cv::imwrite's defaults: the reference
 * encoder's files) through the one-stream kernel's all-offsets turn; the first image's first block decides for the launch. */
int cimbar_hip_png_decode_batch_v(int device, const uint8_t* d_zbuf, size_t zbuf_bytes, const cimbar_hip_png_desc* d_desc, int n, uint8_t* d_scratch,
                                  size_t scratch_stride, uint8_t* d_rgb, size_t rgb_stride, int32_t* d_status, int variant, void* hip_stream);

/* ---- stage taps (parity tests / profiling; all buffers host memory, sized for the LAST decoded batch of n frames) --- */
enum {
	CIMBAR_HIP_TAP_BITPLANE = 0,   /* n * 131072 bytes: CimbReader::_grayscale layout (bit x+1024*y, MSB first) */
	CIMBAR_HIP_TAP_SYMBOLS = 1,    /* n * 12400 bytes : symbol (0..15) by linear cell index */
	CIMBAR_HIP_TAP_COLORS = 2,     /* n * 12400 bytes : colour (0..3) by linear cell index */
	CIMBAR_HIP_TAP_DRIFT = 3,      /* n * 12400 * 2 int8: accumulated (dx,dy) at which each cell's colour is read */
	CIMBAR_HIP_TAP_RS_OK = 4,      /* n * 60 bytes    : 1 = libcorrect-equivalent decode returned > 0, per RS block */
	CIMBAR_HIP_TAP_FLOOD = 5,      /* n bytes         : 1 = frame needed the exact flood-order pass */
	CIMBAR_HIP_TAP_CCM = 6,        /* n * 10 floats   : 3x3 matrix used for the colour pass + active flag */
	CIMBAR_HIP_TAP_FLOOD_PATH = 7, /* n bytes         : 0 = parallel pass was exact, 1 = exact flood replay, 2 = certified batch flood */
	CIMBAR_HIP_TAP_FLOOD_INFO = 8, /* n u32           : what the batch-parallel flood made of a flagged frame: low byte 0 = certified, 1..4 = the rule
	                                  that declined it, 5 = out of super-rounds; bits 8..15 the super-round; bits 16.. cells decoded by then.
	                                  0xFFFFFFFF for frames that were never flagged, and for every frame of a batch in which the pass did not
	                                  run: after a batch where it certified fewer than one in sixteen of the frames it was given, the next fifteen
	                                  batches go straight to the exact replay (CIMBAR_HIP_FLOOD_WAVE_ADAPT=0 at cimbar_hip_create: never skip).
	                                  Which pass produced a frame's symbols never changes them */
	CIMBAR_HIP_TAP_FLOOD_VERIFY = 9 /* n u32          : with CIMBAR_HIP_FLOOD_VERIFY=1 in the environment at cimbar_hip_create, every frame the batch-parallel
	                                  flood certified is ALSO replayed exactly and compared: cells whose symbol or drifted position differed
	                                  (0 = the certificate held; the exact result is what the decode used either way); 0xFFFFFFFF for frames
	                                  that were not certified, or when the mode is off. The context prints the totals to stderr at destroy. */
};
int64_t cimbar_hip_tap(cimbar_hip_ctx* ctx, int what, void* out, size_t out_bytes);

/* Average device time (ms) of each pipeline stage over the last decode_batch call that ran with timing enabled
 * (HIP events on the launch stream). names: static strings. Returns the number of stages written (<= max). */
int cimbar_hip_enable_timing(cimbar_hip_ctx* ctx, int on);
int cimbar_hip_stage_times(cimbar_hip_ctx* ctx, const char** names, float* ms, int max);

#ifdef __cplusplus
}
#endif
#endif /* CIMBAR_HIP_H */
