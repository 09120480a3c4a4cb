/* cimbar_recv_hip.h -- the reference's own receive-side C symbols for the frame-decode step, backed by the MI355X path.
 *
 * libcimbar_recv_hip.so exports, under the reference's names and with its signatures, argument meaning and return codes, the three functions of
 *     /root/reference/src/lib/cimbar_js/cimbar_recv_js.h:16-17,36   (implementation: cimbar_recv_js.cpp:143-189,272-288)
 * that sit on the hot path -- what src/exe/cimbar_recv2/recv2.cpp:109-151 and web/recv-worker.js call per camera frame:
 *
 *     cimbard_configure_decode(mode_val)            selects the configuration (68 B, 67 Bm, 66 Bu, 4, 8; <= 0 and anything unlisted: 68)
 *     cimbard_get_bufsize()                         fountain_chunks_per_frame * fountain_chunk_size of that configuration
 *     cimbard_scan_extract_decode(img, w, h, format, bufspace, bufsize)
 *                                                   get_rgb -> Extractor::extract -> Decoder::decode_fountain into an escrow_buffer_writer:
 *                                                   the capture is scanned, deskewed and decoded on the GPU; the delivered chunks are packed
 *                                                   front to back into bufspace. Returns bytes written (a multiple of the chunk size),
 *                                                   -1 for an empty image, -2 if bufsize < cimbard_get_bufsize(), -3 if no frame was found
 *                                                   -- the reference's codes -- and -4 if the GPU path itself failed (no device, HIP error:
 *                                                   cimbard_get_report then holds the library's message). There is no CPU fallback.
 *
 * Everything behind those calls in the reference's file -- cimbard_fountain_decode (the wirehair sink), cimbard_get_filesize / _get_filename /
 * _decompress_read (zstd) -- is host code outside the decode path and stays the reference's own: a build of the reference links this library
 * in place of those three functions and keeps the rest of cimbar_recv_js.cpp (oracle/Makefile `recvfull` makes exactly
 * that library for tests/test_gpu_recv_shim.py, with the reference file compiled where it lies and its three functions renamed away).
 *
 * State: the selected mode is process-wide, like the reference's `_modeVal`; the decoder behind it (one cimbar_hip context, with its carried
 * colour-correction matrix -- CimbDecoder.cpp:69-73 `static thread_local`) is per calling thread and is re-created when the mode changed.
 */
#ifndef CIMBAR_RECV_HIP_H
#define CIMBAR_RECV_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int cimbard_configure_decode(int mode_val);                                                          /* cimbar_recv_js.h:36 */
int cimbard_get_bufsize(void);                                                                       /* cimbar_recv_js.h:16 */
int cimbard_scan_extract_decode(const unsigned char* imgdata, unsigned imgw, unsigned imgh, int format,
                                unsigned char* bufspace, unsigned bufsize);                          /* cimbar_recv_js.h:17 */
/* cimbar_recv_js.h:11: a short status line about the last call of this thread (the library's error text after a -4) */
unsigned cimbard_get_report(unsigned char* buff, unsigned maxlen);

/* not in the reference: which HIP device the contexts are created on (default 0, or the environment's CIMBAR_HIP_DEVICE at first use) */
int cimbard_hip_set_device(int device);

#ifdef __cplusplus
}
#endif
#endif
