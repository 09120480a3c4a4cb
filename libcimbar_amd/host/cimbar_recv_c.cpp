// cimbar_recv_c.cpp -> libcimbar_recv_hip.so: the reference's receive-side C symbols for the decode step (include/cimbar_recv_hip.h), over the C ABI
// of libcimbar_hip.so. Plain C++ (g++), no HIP headers, no OpenCV. What each function replaces:
//     cimbard_get_bufsize            /root/reference/src/lib/cimbar_js/cimbar_recv_js.cpp:143-146
//     cimbard_scan_extract_decode    cimbar_recv_js.cpp:148-189 (get_rgb :94-120, Extractor::extract, Decoder::decode_fountain, escrow_buffer_writer)
//     cimbard_configure_decode       cimbar_recv_js.cpp:272-288
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/cimbar_hip.h"
#include "../../include/cimbar_recv_hip.h"

#ifdef CIMBARD_HOST_REST
// A build of the reference that keeps the rest of its cimbar_recv_js.cpp (sink, zstd: host code) beside this file: its own configure -- renamed
// at compile time -- still has to run, it resets the sink and updates the thread's cimbar::Config for the functions that stay.
extern "C" int CIMBARD_HOST_REST(int mode_val);
#endif

namespace {

std::atomic<int> g_mode{68};
std::atomic<int> g_device{-1};

int device_ordinal()
{
	int d = g_device.load();
	if (d < 0) {
		const char* v = std::getenv("CIMBAR_HIP_DEVICE");
		d = v ? std::atoi(v) : 0;
		g_device.store(d);
	}
	return d;
}

// one decoder per calling thread, like the reference's thread_local Config + colour-correction matrix
struct ThreadDecoder {
	cimbar_hip_ctx* ctx = nullptr;
	int mode = 0, device = -1;
	std::string report;
	~ThreadDecoder() { if (ctx) cimbar_hip_destroy(ctx); }
	int ensure(int want_mode, int want_device)
	{
		if (ctx && mode == want_mode && device == want_device) return 0;
		if (ctx) { cimbar_hip_destroy(ctx); ctx = nullptr; }
		const int rc = cimbar_hip_create(want_device, want_mode, &ctx);
		if (rc != 0 || !ctx) {
			report = "cimbar_hip_create failed (" + std::to_string(rc) + "): no gfx950 device for the decode path";
			ctx = nullptr;
			return rc ? rc : CIMBAR_HIP_EHIP;
		}
		mode = want_mode;
		device = want_device;
		return 0;
	}
};
thread_local ThreadDecoder t_dec;

}  // namespace

extern "C" {

int cimbard_hip_set_device(int device)
{
	if (device < 0) return -1;
	g_device.store(device);
	return 0;
}

int cimbard_configure_decode(int mode_val)
{
	if (mode_val <= 0) mode_val = 68;          // (cimbar_recv_js.cpp:274-276)
	g_mode.store(mode_val);                    // the thread's context follows at its next frame; the sink reset is the host build's business
#ifdef CIMBARD_HOST_REST
	return CIMBARD_HOST_REST(mode_val);
#else
	return 0;
#endif
}

int cimbard_get_bufsize(void)
{
	return cimbar_hip_mode_bufsize(g_mode.load());
}

unsigned cimbard_get_report(unsigned char* buff, unsigned maxlen)
{
	const std::string& r = t_dec.report;
	const unsigned len = r.size() < maxlen ? (unsigned)r.size() : maxlen;
	if (len && buff) std::memcpy(buff, r.data(), len);
	return len;
}

int cimbard_scan_extract_decode(const unsigned char* imgdata, unsigned imgw, unsigned imgh, int format, unsigned char* bufspace, unsigned bufsize)
{
	if (format <= 0) format = 3;
	if (imgw == 0 || imgh == 0) return -1;
	const int mode = g_mode.load();
	const int frame_bytes = cimbar_hip_mode_bufsize(mode);
	if (bufsize < (unsigned)frame_bytes) return -2;          // (cimbar_recv_js.cpp:158-159)
	if (!imgdata || !bufspace) { t_dec.report = "null buffer"; return -4; }
	if (t_dec.ensure(mode, device_ordinal()) != 0) return -4;
	int32_t geo[CIMBAR_HIP_GEOMETRY_WORDS];
	if (cimbar_hip_geometry(t_dec.ctx, geo) != CIMBAR_HIP_GEOMETRY_WORDS) { t_dec.report = "geometry"; return -4; }
	const unsigned per = (unsigned)geo[4], cs = (unsigned)geo[5];
	uint8_t chunks[CIMBAR_HIP_MAX_FRAME_BYTES];
	uint32_t mask = 0;
	int status = 0;
	// The capture crosses to the device as the camera delivered it (`format` is passed through: the conversion get_rgb does on the host happens inside
	// the kernels that read the capture), extract and decode follow without leaving the device. Upstream always sharpens on this path (:166) and
	// uses the header-derived colour correction (Decoder::decode_fountain's default 2).
	const int64_t rc = cimbar_hip_scan_extract_decode_batch_fmt(t_dec.ctx, imgdata, imgw, imgh, format, 1, CIMBAR_HIP_MEM_HOST, /*preprocess*/ 1,
	                                                            /*color_correction*/ 2, chunks, &mask, &status, CIMBAR_HIP_MEM_HOST, nullptr);
	if (rc < 0) {
		t_dec.report = std::string("cimbar_hip_scan_extract_decode_batch_fmt failed (") + std::to_string((long long)rc) + "): " + cimbar_hip_last_error(t_dec.ctx);
		return -4;
	}
	if (status <= 0) { t_dec.report = "no frame found"; return -3; }          // Extractor::FAILURE (:170-171)
	unsigned used = 0;                                                        // escrow_buffer_writer: the delivered chunks, packed front to back
	for (unsigned j = 0; j < per; ++j)
		if (mask & (1u << j)) std::memcpy(bufspace + (size_t)cs * used++, chunks + (size_t)cs * j, cs);
	t_dec.report = "decoded " + std::to_string(used * cs) + " bytes" + (status == 2 ? " (needs sharpen)" : "");
	return (int)(used * cs);
}

}  // extern "C"
