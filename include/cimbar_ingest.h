/* cimbar_ingest.h -- host-side ingest for the MI355X cimbar decode path: what stands between image FILES and the
 * device-resident frames include/cimbar_hip.h wants. It replaces, for that step only, the front of ./cimbar's decode loop
 *     cv::imread(path) + cv::cvtColor(BGR2RGB)        /root/reference/src/exe/cimbar/cimbar.cpp:132-133 (loop :124-162)
 * with a pool of PNG-decoding host threads feeding a ring of pinned batches, whose host->device copies run on a copy stream and overlap
 * the decode of the batch before (cimbar_hip_decode_batch_pipelined). Plain C ABI, C++ inside (libcimbar_ingest.so links zlib, pthreads
 * and libcimbar_hip.so). PNG (1..16-bit gray, RGB, RGBA, gray + alpha, palette; Adam7-interlaced or not) and baseline JPEG (sequential Huffman,
 * 8 bits, gray or YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart intervals; libjpeg's arithmetic, so the pixels are what cv::imread returns) -- what the
 * reference's encoder writes and what its sample set holds (src/lib/encoder/test/DecoderTest.cpp:60,104). */
#ifndef CIMBAR_INGEST_H
#define CIMBAR_INGEST_H

#include <stddef.h>
#include <stdint.h>

#include "cimbar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
	CIMBAR_INGEST_EFORMAT = -20,   /* not a PNG / JPEG these decoders handle (unknown colour type, progressive JPEG, corrupt stream) */
	CIMBAR_INGEST_EIO = -21,       /* a file could not be read */
	CIMBAR_INGEST_ESIZE = -22      /* an image is not image_size_x x image_size_y of the context's mode (frames must be deskewed already: the --no-deskew path, cimbar.cpp:136) */
};

/* One PNG in memory -> tightly packed RGB8 (what cv::imread + BGR2RGB hands the decoder: alpha dropped, gray replicated).
 * rgb may be NULL to query the size: *width / *height are always set. Returns 0 or a negative code. */
int cimbar_png_decode(const uint8_t* png, size_t len, uint8_t* rgb, size_t rgb_capacity, unsigned* width, unsigned* height);

/* One baseline JPEG in memory -> RGB8, with libjpeg(-turbo)'s default arithmetic (ISLOW inverse DCT, "fancy" chroma upsampling, the 16-bit
 * YCbCr tables): what cv::imread + BGR2RGB returns. Progressive / arithmetic-coded / 12-bit / CMYK files: CIMBAR_INGEST_EFORMAT. */
int cimbar_jpeg_decode(const uint8_t* jpg, size_t len, uint8_t* rgb, size_t rgb_capacity, unsigned* width, unsigned* height);
/* either of the two, by the file's first bytes */
int cimbar_image_decode(const uint8_t* file, size_t len, uint8_t* rgb, size_t rgb_capacity, unsigned* width, unsigned* height);

typedef struct cimbar_ingest cimbar_ingest;

/* called once per batch, in frame order, from the thread that called cimbar_ingest_run_*: chunks = n * 7500 bytes (mode B; in general n *
 * geometry[4] * geometry[5] of cimbar_hip_geometry), masks = n words (host
 * memory, valid during the call); first_frame = index of the batch's first frame in the input list. Return non-zero to stop early
 * (e.g. the fountain sink is complete). */
typedef int (*cimbar_ingest_sink_fn)(void* user, const uint8_t* chunks, const uint32_t* masks, int first_frame, int n);

/* threads: PNG decode threads (<= 0: one per CPU the process may use -- hardware threads or the cgroup CPU quota, whichever is smaller -- at most 128); batch_frames: frames per device batch (<= 0: 64);
 * ring: batches in flight between the host pool and the device (2..4, <= 0: 3) */
int cimbar_ingest_create(cimbar_hip_ctx* ctx, int threads, int batch_frames, int ring, cimbar_ingest** out);

/* Where cimbar_ingest_run_files decodes the PNGs.
 *   CIMBAR_INGEST_PNG_HOST   : on the host pool (zlib inflate + un-filter per thread), decoded frames cross PCIe -- cimbar_ingest_create.
 *   CIMBAR_INGEST_PNG_DEVICE : the host threads only read the files and copy their IDAT payloads into the pinned batch; the compressed bytes
 *       cross PCIe and cimbar_hip_png_decode_batch (include/cimbar_hip.h) inflates and un-filters them on the device, straight into the
 *       frames the decoder reads. Takes what cv::imwrite / any ordinary tool writes (8 bits per sample, gray / RGB / RGBA / palette, not
 *       interlaced). A file outside that -- a JPEG, a 16-bit / sub-byte / Adam7 PNG -- is decoded by the host thread that read it and its frame
 *       joins the batch on the device (up to 32 such files per batch; cimbar_ingest_host_decoded counts them); a PNG whose stream the device
 *       refuses (invalid deflate data, Adler-32 mismatch) is skipped like an unreadable one. batch_frames <= 0: 512 (the inflate pass runs one wavefront per image:
 *       large batches are what fills the GPU); zbytes_per_frame: pinned + device room for a frame's compressed stream (0: a quarter of the
 *       decoded frame; a batch whose streams together exceed batch_frames * zbytes_per_frame loses the files that no longer fit).
 *       cimbar_ingest_run_raw is not available on such an ingest. */
enum { CIMBAR_INGEST_PNG_HOST = 0, CIMBAR_INGEST_PNG_DEVICE = 1 };
int cimbar_ingest_create_ex(cimbar_hip_ctx* ctx, int threads, int batch_frames, int ring, int png_mode, size_t zbytes_per_frame, cimbar_ingest** out);
/* device PNG mode, last cimbar_ingest_run_files: [0] files seen, [1] refused by the host's chunk walk (or unreadable / wrong size / no room),
 * [2] refused by the device, [3] bytes copied to the device */
int cimbar_ingest_png_stats(const cimbar_ingest* ing, int64_t out4[4]);
/* device PNG mode, last cimbar_ingest_run_files: files decoded on the host instead (JPEG, PNGs the kernels do not take) */
int64_t cimbar_ingest_host_decoded(const cimbar_ingest* ing);
/* device PNG mode, last cimbar_ingest_run_files: readable, right-sized files that needed the host decoder AFTER their batch's 32 fallback frames
 * were taken and were therefore dropped (cv::imread would have decoded them). They are also part of png_stats [1], but unlike an unreadable
 * file they are a capacity limit of this library: a caller that feeds directories of JPEGs / 16-bit PNGs should use a host-mode ingest, and
 * one that sees a non-zero count here has lost frames. Not handled at all: EXIF orientation (cv::imread rotates by it; csrc/jpeg.inc does not). */
int64_t cimbar_ingest_fallback_overflow(const cimbar_ingest* ing);
void cimbar_ingest_destroy(cimbar_ingest* ing);
const char* cimbar_ingest_last_error(const cimbar_ingest* ing);

/* ./cimbar --no-deskew img1.png img2.png ... (cimbar.cpp:124-162, fountain mode): every file is a deskewed frame of the context's mode (1024x1024 in mode B,
 * 1024x720 in mode 67). Files that cannot be read or are not PNGs of that size are skipped like the reference skips frames it cannot decode (their slots deliver nothing).
 * Returns the total good bytes (sum of what Decoder::decode_fountain would have returned per frame) or a negative code. */
int64_t cimbar_ingest_run_files(cimbar_ingest* ing, const char* const* paths, int nfiles, int should_preprocess, int color_correction,
                                cimbar_ingest_sink_fn sink, void* user);

/* The same pipeline for frames that are already raw RGB8 in host memory: pageable memory is staged into the pinned ring by the host threads;
 * page-locked memory (hipHostMalloc / hipHostRegister) is copied to the device where it lies. H2D on the copy stream, decode and D2H all
 * overlapped. */
int64_t cimbar_ingest_run_raw(cimbar_ingest* ing, const uint8_t* frames, int n, int should_preprocess, int color_correction,
                              cimbar_ingest_sink_fn sink, void* user);

/* seconds spent in the last run: [0] wall, [1] PNG decode / staging (summed over threads), [2] waiting for the device */
int cimbar_ingest_timings(const cimbar_ingest* ing, double out3[3]);

#ifdef __cplusplus
}
#endif
#endif /* CIMBAR_INGEST_H */
