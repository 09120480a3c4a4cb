// ingest.cpp -> libcimbar_ingest.so: the host side in front of the device decode path (include/cimbar_ingest.h).
//
//   files / host frames --[pool of host threads: read + PNG decode (zlib inflate + un-filter) | staging copy]--> pinned batch (ring of R)
//        --[copy stream: hipMemcpyAsync H2D]--> device batch --[context streams: cimbar_hip_decode_batch_pipelined]--> chunks, masks
//        --[output stream: D2H into pinned memory]--> sink callback, in frame order
//
// The batch being filled, the batch being copied and the batches being decoded are different ring slots, so PNG decoding, PCIe traffic and
// the kernels overlap; the calling thread only issues work and hands finished batches to the callback. It replaces cv::imread + cvtColor of
// ./cimbar's loop (/root/reference/src/exe/cimbar/cimbar.cpp:124-162), which decodes one file at a time on one thread.
//
// Device PNG mode (cimbar_ingest_create_ex(..., CIMBAR_INGEST_PNG_DEVICE)): the host threads only read the files and copy their IDAT payloads
// into the pinned batch; the compressed bytes cross PCIe (a tenth of the decoded frames) and cimbar_hip_png_decode_batch inflates and
// un-filters them on the device, one stream per ring slot so that the inflate passes of consecutive batches overlap:
//   files --[pool: read + chunk walk + memcpy of the IDAT bytes]--> pinned zlib streams + descriptors --[copy stream]--> device
//        --[slot stream: k_png_inflate, k_png_unfilter]--> device frames --[context streams: decode]--> chunks, masks, PNG status --> sink
#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cimbar_ingest.h"

namespace {


// ---------------------------------------------------------------------------------------------------------------- PNG
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
	const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// PNG (ISO/IEC 15948): signature, IHDR, [PLTE], IDAT..., IEND; zlib stream of filtered scanlines. Output: RGB8, alpha dropped, gray
// replicated, 16-bit samples reduced to their high byte, Adam7 passes put back together -- what cv::imread(IMREAD_COLOR) followed by BGR2RGB
// gives the reference.
int png_decode(const uint8_t* png, size_t len, uint8_t* rgb, size_t cap, unsigned* pw, unsigned* ph, std::vector<uint8_t>& idat, std::vector<uint8_t>& raw)
{
	static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	if (len < 8 + 25 || std::memcmp(png, SIG, 8) != 0) return CIMBAR_INGEST_EFORMAT;
	size_t pos = 8;
	unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
	uint8_t pal[256][3];
	unsigned npal = 0;
	bool have_ihdr = false;
	idat.clear();
	while (pos + 12 <= len) {
		const uint32_t clen = be32(png + pos);
		const uint8_t* tag = png + pos + 4;
		const uint8_t* data = png + pos + 8;
		if ((size_t)clen > len - pos - 12) return CIMBAR_INGEST_EFORMAT;
		if (!std::memcmp(tag, "IHDR", 4)) {
			if (clen != 13) return CIMBAR_INGEST_EFORMAT;
			w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
			have_ihdr = true;
		} else if (!std::memcmp(tag, "PLTE", 4)) {
			npal = clen / 3 > 256 ? 256 : clen / 3;
			for (unsigned i = 0; i < npal; ++i) { pal[i][0] = data[3 * i]; pal[i][1] = data[3 * i + 1]; pal[i][2] = data[3 * i + 2]; }
		} else if (!std::memcmp(tag, "IDAT", 4)) {
			idat.insert(idat.end(), data, data + clen);
		} else if (!std::memcmp(tag, "IEND", 4)) {
			break;
		}
		pos += 12 + (size_t)clen;
	}
	if (!have_ihdr || w == 0 || h == 0 || w > 32768 || h > 32768) return CIMBAR_INGEST_EFORMAT;
	if (pw) *pw = w;
	if (ph) *ph = h;
	if (interlace > 1) return CIMBAR_INGEST_EFORMAT;
	unsigned channels;
	switch (ctype) {
		case 0: channels = 1; break;
		case 2: channels = 3; break;
		case 3: channels = 1; break;
		case 4: channels = 2; break;
		case 6: channels = 4; break;
		default: return CIMBAR_INGEST_EFORMAT;
	}
	if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return CIMBAR_INGEST_EFORMAT;
	if (ctype == 3 && depth == 16) return CIMBAR_INGEST_EFORMAT;
	if (!rgb) return 0;
	if (cap < (size_t)w * h * 3) return CIMBAR_HIP_EINVAL;
	const size_t bits_pp = (size_t)channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
	// the reduced images the stream holds one after the other: the whole image, or Adam7's seven passes (ISO/IEC 15948 8.2):
	// pass p covers pixels (x0 + i * dx, y0 + j * dy)
	struct Pass { unsigned x0, y0, dx, dy; };
	static const Pass ADAM7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
	static const Pass WHOLE = {0, 0, 1, 1};
	const int npass = interlace ? 7 : 1;
	size_t total = 0;
	for (int p = 0; p < npass; ++p) {
		const Pass& ps = interlace ? ADAM7[p] : WHOLE;
		const size_t pw_ = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph_ = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
		if (pw_ && ph_) total += ((pw_ * bits_pp + 7) / 8 + 1) * ph_;
	}
	raw.resize(total);
	uLongf out_len = (uLongf)raw.size();
	if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return CIMBAR_INGEST_EFORMAT;
	size_t base = 0;
	for (int p = 0; p < npass; ++p) {
		const Pass& ps = interlace ? ADAM7[p] : WHOLE;
		const size_t pw_ = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph_ = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
		if (!pw_ || !ph_) continue;
		const size_t row_bytes = (pw_ * bits_pp + 7) / 8;
		// un-filter in place (filter type byte in front of every scanline)
		for (size_t y = 0; y < ph_; ++y) {
			uint8_t* cur = raw.data() + base + (row_bytes + 1) * y + 1;
			const uint8_t* up = y ? cur - (row_bytes + 1) : nullptr;
			switch (cur[-1]) {
				case 0: break;
				case 1: for (size_t i = bpp; i < row_bytes; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
				case 2: if (up) for (size_t i = 0; i < row_bytes; ++i) cur[i] = (uint8_t)(cur[i] + up[i]); break;
				case 3:
					for (size_t i = 0; i < row_bytes; ++i) {
						const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
						cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
					}
					break;
				case 4:
					for (size_t i = 0; i < row_bytes; ++i) {
						const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
						cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
					}
					break;
				default: return CIMBAR_INGEST_EFORMAT;
			}
			const size_t oy = ps.y0 + y * ps.dy;
			if (!interlace && depth == 8 && ctype == 2) { std::memcpy(rgb + oy * w * 3, cur, (size_t)w * 3); continue; }
			for (size_t x = 0; x < pw_; ++x) {
				unsigned smp[4] = {0, 0, 0, 0};
				if (depth == 8) for (unsigned c = 0; c < channels; ++c) smp[c] = cur[x * channels + c];
				else if (depth == 16) for (unsigned c = 0; c < channels; ++c) smp[c] = cur[(x * channels + c) * 2];
				else {
					const unsigned per = 8 / depth, v = (cur[x / per] >> ((per - 1 - x % per) * depth)) & ((1u << depth) - 1u);
					smp[0] = ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
				}
				uint8_t* dst = rgb + (oy * w + ps.x0 + x * ps.dx) * 3;
				if (ctype == 3) { const unsigned i = smp[0] < npal ? smp[0] : 0; dst[0] = pal[i][0]; dst[1] = pal[i][1]; dst[2] = pal[i][2]; }
				else if (ctype == 0 || ctype == 4) { dst[0] = dst[1] = dst[2] = (uint8_t)smp[0]; }
				else { dst[0] = (uint8_t)smp[0]; dst[1] = (uint8_t)smp[1]; dst[2] = (uint8_t)smp[2]; }
			}
		}
		base += (row_bytes + 1) * ph_;
	}
	return 0;
}

// device PNG mode: walk the chunks, return the geometry and where the IDAT payloads are; nothing is decompressed on the host.
// Only what the device kernels take (8 bits per sample, non-interlaced, colour type 0 / 2 / 3 / 6) is accepted.
struct PngInfo { unsigned w = 0, h = 0, ctype = 0; size_t zlen = 0; const uint8_t* plte = nullptr; unsigned npal = 0; std::vector<std::pair<const uint8_t*, uint32_t>> idat; };
int png_walk(const uint8_t* png, size_t len, PngInfo& info)
{
	static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	if (len < 8 + 25 || std::memcmp(png, SIG, 8) != 0) return CIMBAR_INGEST_EFORMAT;
	size_t pos = 8;
	unsigned depth = 0, interlace = 0;
	bool have_ihdr = false;
	info.idat.clear(); info.zlen = 0; info.plte = nullptr; info.npal = 0;
	while (pos + 12 <= len) {
		const uint32_t clen = be32(png + pos);
		const uint8_t* tag = png + pos + 4;
		const uint8_t* data = png + pos + 8;
		if ((size_t)clen > len - pos - 12) return CIMBAR_INGEST_EFORMAT;
		if (!std::memcmp(tag, "IHDR", 4)) {
			if (clen != 13) return CIMBAR_INGEST_EFORMAT;
			info.w = be32(data); info.h = be32(data + 4); depth = data[8]; info.ctype = data[9]; interlace = data[12];
			have_ihdr = true;
		} else if (!std::memcmp(tag, "PLTE", 4)) { info.plte = data; info.npal = clen / 3 > 256 ? 256 : clen / 3; }
		else if (!std::memcmp(tag, "IDAT", 4)) { if (clen) info.idat.emplace_back(data, clen); info.zlen += clen; }
		else if (!std::memcmp(tag, "IEND", 4)) break;
		pos += 12 + (size_t)clen;
	}
	if (!have_ihdr || info.w == 0 || info.h == 0 || depth != 8 || interlace != 0) return CIMBAR_INGEST_EFORMAT;
	if (!(info.ctype == 0 || info.ctype == 2 || info.ctype == 3 || info.ctype == 6) || info.zlen < 6 || info.zlen >= (1u << 28)) return CIMBAR_INGEST_EFORMAT;   // the device kernels' bit positions are 32-bit: streams below 256 MiB (cimbar_hip.h)
	if (info.ctype == 3 && !info.plte) return CIMBAR_INGEST_EFORMAT;
	return 0;
}

#include "jpeg.inc"

// what cv::imread(path) + cvtColor(BGR2RGB) makes of a file (cimbar.cpp:132-133), by its magic bytes: PNG or (baseline) JPEG
int image_decode(const uint8_t* file, size_t len, uint8_t* rgb, size_t cap, unsigned* pw, unsigned* ph, std::vector<uint8_t>& idat, std::vector<uint8_t>& raw)
{
	if (len >= 2 && file[0] == 0xFF && file[1] == 0xD8) return jpeg::decode(file, len, rgb, cap, pw, ph);
	return png_decode(file, len, rgb, cap, pw, ph, idat, raw);
}

bool read_file(const char* path, std::vector<uint8_t>& out)
{
	FILE* f = std::fopen(path, "rb");
	if (!f) return false;
	std::fseek(f, 0, SEEK_END);
	const long n = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	if (n <= 0) { std::fclose(f); return false; }
	out.resize((size_t)n);
	const bool ok = std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
	std::fclose(f);
	return ok;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- pipeline
struct cimbar_ingest {
	unsigned fw = 0, fh = 0;          // image_size_x / image_size_y of the context's mode (cimbar_hip_geometry)
	size_t frame = 0, frame_bytes = 0; // RGB8 bytes per frame; chunk bytes per frame (chunks per frame * chunk size)
	unsigned chunk = 0;
	cimbar_hip_ctx* ctx = nullptr;
	int device = 0, threads = 1, B = 64, R = 3;
	std::string err;
	struct Slot {
		uint8_t* h_in = nullptr; uint8_t* d_in = nullptr;
		uint8_t* h_chunks = nullptr; uint8_t* d_chunks = nullptr;
		uint32_t* h_masks = nullptr; uint32_t* d_masks = nullptr;
		hipEvent_t done = nullptr;
		std::vector<uint8_t> valid;
		// device PNG mode
		uint8_t* h_z = nullptr; uint8_t* d_z = nullptr;                       // zlib streams (+ palettes) of the batch, 16-byte aligned pieces
		cimbar_hip_png_desc* h_desc = nullptr; cimbar_hip_png_desc* d_desc = nullptr;
		uint8_t* d_scratch = nullptr;                                           // filtered scanlines between the two PNG kernels
		int32_t* h_status = nullptr; int32_t* d_status = nullptr;
		hipStream_t png_stream = nullptr;
		hipEvent_t copied = nullptr;
		// ... and the files the device kernels do not take (JPEG; 16-bit, sub-byte, Adam7 PNGs): decoded by the host thread into one of FB
		// pinned frames and copied into the batch behind the PNG kernels
		uint8_t* h_fb = nullptr;
		std::vector<int> fb_index;                                              // per frame of the batch: its pinned fallback frame, or -1
	};
	static constexpr int FB = 32;     // host-decoded frames a device-mode batch can take
	int png_mode = CIMBAR_INGEST_PNG_HOST;
	size_t zcap = 0;                  // bytes of h_z / d_z per slot
	size_t scratch_stride = 0;
	std::vector<Slot> slots;
	hipStream_t copy_stream = nullptr, out_stream = nullptr;
	double t_wall = 0, t_host = 0, t_wait = 0;
	int64_t png_files = 0, png_refused_host = 0, png_refused_device = 0, png_bytes = 0, host_decoded = 0;   // device PNG mode, last run
	std::atomic<int64_t> fb_overflow{0};   // ... files that needed a host-decoded fallback frame when the batch's FB of them were taken (dropped; counted apart from unreadable files)
};

namespace {

#define ICHK(call)                                                                       \
	do {                                                                                 \
		hipError_t e__ = (call);                                                         \
		if (e__ != hipSuccess) {                                                         \
			ing->err = std::string(#call) + ": " + hipGetErrorString(e__);              \
			return CIMBAR_HIP_EHIP;                                                      \
		}                                                                                \
	} while (0)

// fill(i, slot, j, zcursor) produces frame i as entry j of the slot's batch (host mode: RGB8 into the pinned frames; device PNG mode: the
// file's zlib stream + descriptor, the space taken from the batch's cursor) and says whether it is a usable frame
// direct != nullptr: the frames already sit in page-locked host memory (hipHostMalloc / hipHostRegister) -- no staging threads, the H2D
// copies read the caller's buffer
constexpr size_t ZMASK = ((size_t)1 << 48) - 1, FB_ONE = (size_t)1 << 48;

template <typename FILL>
int64_t run_pipeline(cimbar_ingest* ing, int n, int pre, int cc, cimbar_ingest_sink_fn sink, void* user, FILL fill, const uint8_t* direct = nullptr)
{
	if (n <= 0) return 0;
	ICHK(hipSetDevice(ing->device));
	const int B = ing->B, R = ing->R, nbatch = (n + B - 1) / B;
	const bool dev_png = ing->png_mode == CIMBAR_INGEST_PNG_DEVICE;
	// per batch: bytes of compressed streams handed out (low 48 bits) and host-decoded fallback frames handed out (the bits above)
	std::unique_ptr<std::atomic<size_t>[]> zcur(new std::atomic<size_t>[(size_t)nbatch]);
	for (int k = 0; k < nbatch; ++k) zcur[(size_t)k].store(0);
	ing->png_files = ing->png_refused_host = ing->png_refused_device = ing->png_bytes = ing->host_decoded = 0;
	ing->fb_overflow = 0;
	std::mutex mu;
	std::condition_variable cv;
	std::vector<int> filled((size_t)nbatch, 0);      // frames of batch k staged so far
	std::vector<int> freed((size_t)R, 0);            // how many times slot s has been handed back
	std::atomic<int> next{0};
	std::atomic<bool> stop{false};
	double host_s = 0;

	auto worker = [&]() {
		double mine = 0;
		for (;;) {
			const int i = next.fetch_add(1);
			if (i >= n || stop.load()) break;
			const int k = i / B, s = k % R;
			{
				std::unique_lock<std::mutex> lk(mu);     // the slot's previous batch (k - R) must have been consumed
				cv.wait(lk, [&] { return freed[s] >= k / R || stop.load(); });
			}
			if (stop.load()) break;
			const double t0 = now_s();
			const bool ok = fill(i, ing->slots[s], i - k * B, zcur[(size_t)k]);
			ing->slots[s].valid[(size_t)(i - k * B)] = ok ? 1 : 0;
			mine += now_s() - t0;
			{
				std::lock_guard<std::mutex> lk(mu);
				filled[(size_t)k] += 1;
			}
			cv.notify_all();
		}
		std::lock_guard<std::mutex> lk(mu);
		host_s += mine;
	};
	std::vector<std::thread> pool;
	const int nthreads = direct ? 0 : (ing->threads < n ? ing->threads : n);
	for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
	if (direct) {
		for (int k = 0; k < nbatch; ++k) filled[(size_t)k] = (k == nbatch - 1) ? n - k * B : B;
		for (auto& sl : ing->slots) std::fill(sl.valid.begin(), sl.valid.end(), (uint8_t)1);
	}

	int64_t total = 0;
	int rc = 0;
	double wait_s = 0;
	const double t_start = now_s();
	auto consume = [&](int k) -> int {
		cimbar_ingest::Slot& sl = ing->slots[(size_t)(k % R)];
		const int m = (k == nbatch - 1) ? n - k * B : B;
		const double t0 = now_s();
		hipError_t e = hipEventSynchronize(sl.done);
		wait_s += now_s() - t0;
		if (e != hipSuccess) { ing->err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); return CIMBAR_HIP_EHIP; }
		for (int j = 0; j < m; ++j) {
			if (dev_png) {
				ing->png_files += 1;
				if (!sl.valid[(size_t)j]) ing->png_refused_host += 1;
				else if (sl.fb_index[(size_t)j] >= 0) ing->host_decoded += 1;            // (its descriptor was empty: the device's verdict on it means nothing)
				else if (sl.h_status[j] != 0) { ing->png_refused_device += 1; sl.valid[(size_t)j] = 0; }
			}
			if (!sl.valid[(size_t)j]) { sl.h_masks[j] = 0; std::memset(sl.h_chunks + (size_t)j * ing->frame_bytes, 0, ing->frame_bytes); }
			total += (int64_t)ing->chunk * __builtin_popcount(sl.h_masks[j] & 0xFFFu);
		}
		const int stop_now = sink ? sink(user, sl.h_chunks, sl.h_masks, k * B, m) : 0;
		{
			std::lock_guard<std::mutex> lk(mu);
			freed[(size_t)(k % R)] += 1;
		}
		cv.notify_all();
		return stop_now ? 1 : 0;
	};
	int consumed = 0;
	for (int k = 0; k < nbatch && rc == 0; ++k) {
		cimbar_ingest::Slot& sl = ing->slots[(size_t)(k % R)];
		const int m = (k == nbatch - 1) ? n - k * B : B;
		{
			const double t0 = now_s();
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return filled[(size_t)k] == m; });
			(void)t0;
		}
		int r;
		if (dev_png) {
			// the compressed bytes and the descriptors go over on the copy stream; the slot's own stream inflates and un-filters them into
			// d_in (so that consecutive batches' inflate passes overlap) and is what the decoder then waits for
			const size_t zbytes = ((zcur[(size_t)k].load() & ZMASK) + 15) & ~(size_t)15;
			ing->png_bytes += (int64_t)zbytes;
			hipError_t e = zbytes ? hipMemcpyAsync(sl.d_z, sl.h_z, zbytes < ing->zcap ? zbytes : ing->zcap, hipMemcpyHostToDevice, ing->copy_stream) : hipSuccess;
			if (e == hipSuccess) e = hipMemcpyAsync(sl.d_desc, sl.h_desc, sizeof(cimbar_hip_png_desc) * (size_t)m, hipMemcpyHostToDevice, ing->copy_stream);
			if (e == hipSuccess) e = hipEventRecord(sl.copied, ing->copy_stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(sl.png_stream, sl.copied, 0);
			if (e != hipSuccess) { ing->err = std::string("copy of the PNG streams: ") + hipGetErrorString(e); rc = CIMBAR_HIP_EHIP; break; }
			// (R batches are kept in flight: with 8192+ images among them the four-streams-per-wavefront inflate is the faster one)
			r = cimbar_hip_png_decode_batch_v(ing->device, sl.d_z, ing->zcap, sl.d_desc, m, sl.d_scratch, ing->scratch_stride, sl.d_in, ing->frame, sl.d_status,
			                                  (long)B * R >= 8192 ? 4 : 0, sl.png_stream);
			if (r != 0) { ing->err = "cimbar_hip_png_decode_batch failed to launch"; rc = r; break; }
			for (int j = 0; j < m && e == hipSuccess; ++j)                               // host-decoded frames join the batch behind the PNG kernels
				if (sl.valid[(size_t)j] && sl.fb_index[(size_t)j] >= 0)
					e = hipMemcpyAsync(sl.d_in + (size_t)j * ing->frame, sl.h_fb + (size_t)sl.fb_index[(size_t)j] * ing->frame, ing->frame, hipMemcpyHostToDevice, sl.png_stream);
			if (e != hipSuccess) { ing->err = std::string("copy of a host-decoded frame: ") + hipGetErrorString(e); rc = CIMBAR_HIP_EHIP; break; }
			r = cimbar_hip_decode_batch_pipelined(ing->ctx, sl.d_in, m, pre, cc, sl.d_chunks, sl.d_masks, sl.png_stream);
		} else {
			hipError_t e = hipMemcpyAsync(sl.d_in, direct ? direct + (size_t)k * B * ing->frame : sl.h_in, (size_t)m * ing->frame, hipMemcpyHostToDevice, ing->copy_stream);
			if (e != hipSuccess) { ing->err = std::string("hipMemcpyAsync: ") + hipGetErrorString(e); rc = CIMBAR_HIP_EHIP; break; }
			// "the frames are whatever hip_stream has produced up to here": the library's own stream waits for the copy
			r = cimbar_hip_decode_batch_pipelined(ing->ctx, sl.d_in, m, pre, cc, sl.d_chunks, sl.d_masks, ing->copy_stream);
		}
		if (r == 0) r = cimbar_hip_pipeline_wait(ing->ctx, ing->out_stream, 0);
		if (r != 0) { ing->err = std::string("decode: ") + cimbar_hip_last_error(ing->ctx); rc = r; break; }
		if (dev_png) (void)hipMemcpyAsync(sl.h_status, sl.d_status, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, ing->out_stream);
		(void)hipMemcpyAsync(sl.h_chunks, sl.d_chunks, (size_t)m * ing->frame_bytes, hipMemcpyDeviceToHost, ing->out_stream);
		(void)hipMemcpyAsync(sl.h_masks, sl.d_masks, sizeof(uint32_t) * (size_t)m, hipMemcpyDeviceToHost, ing->out_stream);
		(void)hipEventRecord(sl.done, ing->out_stream);
		// keep R - 1 batches in flight behind the one just issued
		while (consumed <= k - (R - 1) && rc == 0) {
			const int c = consume(consumed++);
			if (c < 0) rc = c;
			else if (c > 0) { rc = 1; }
		}
	}
	while (consumed < nbatch && rc == 0) {
		// (only batches that were issued can be consumed; an early stop leaves the rest)
		const int c = consume(consumed++);
		if (c < 0) rc = c;
		else if (c > 0) rc = 1;
	}
	{
		// under the mutex: a worker that has just found its wait predicate false must not miss this wake-up (it would sleep for ever and join() hang)
		std::lock_guard<std::mutex> lk(mu);
		stop.store(true);
	}
	cv.notify_all();
	for (auto& t : pool) t.join();
	(void)hipStreamSynchronize(ing->copy_stream);
	for (auto& sl : ing->slots) if (sl.png_stream) (void)hipStreamSynchronize(sl.png_stream);
	(void)hipStreamSynchronize(ing->out_stream);
	(void)cimbar_hip_pipeline_wait(ing->ctx, ing->out_stream, 0);
	(void)hipStreamSynchronize(ing->out_stream);
	ing->t_wall = now_s() - t_start;
	ing->t_host = host_s;
	ing->t_wait = wait_s;
	return rc < 0 ? rc : total;
}

}  // namespace

extern "C" {

int cimbar_png_decode(const uint8_t* png, size_t len, uint8_t* rgb, size_t rgb_capacity, unsigned* width, unsigned* height)
{
	if (!png) return CIMBAR_HIP_EINVAL;
	std::vector<uint8_t> idat, raw;
	return png_decode(png, len, rgb, rgb_capacity, width, height, idat, raw);
}

int cimbar_jpeg_decode(const uint8_t* jpg, size_t len, uint8_t* rgb, size_t rgb_capacity, unsigned* width, unsigned* height)
{
	if (!jpg) return CIMBAR_HIP_EINVAL;
	return jpeg::decode(jpg, len, rgb, rgb_capacity, width, height);
}

int cimbar_image_decode(const uint8_t* file, size_t len, uint8_t* rgb, size_t rgb_capacity, unsigned* width, unsigned* height)
{
	if (!file) return CIMBAR_HIP_EINVAL;
	std::vector<uint8_t> idat, raw;
	return image_decode(file, len, rgb, rgb_capacity, width, height, idat, raw);
}

int64_t cimbar_ingest_host_decoded(const cimbar_ingest* ing) { return ing ? ing->host_decoded : CIMBAR_HIP_EINVAL; }
int64_t cimbar_ingest_fallback_overflow(const cimbar_ingest* ing) { return ing ? ing->fb_overflow.load() : CIMBAR_HIP_EINVAL; }

int cimbar_ingest_create(cimbar_hip_ctx* ctx, int threads, int batch_frames, int ring, cimbar_ingest** out)
{
	return cimbar_ingest_create_ex(ctx, threads, batch_frames, ring, CIMBAR_INGEST_PNG_HOST, 0, out);
}

int cimbar_ingest_create_ex(cimbar_hip_ctx* ctx, int threads, int batch_frames, int ring, int png_mode, size_t zbytes_per_frame, cimbar_ingest** out)
{
	if (!ctx || !out || !(png_mode == CIMBAR_INGEST_PNG_HOST || png_mode == CIMBAR_INGEST_PNG_DEVICE)) return CIMBAR_HIP_EINVAL;
	*out = nullptr;
	cimbar_ingest* ing = new cimbar_ingest();
	ing->ctx = ctx;
	ing->png_mode = png_mode;
	ing->device = cimbar_hip_device(ctx);
	int32_t geo[CIMBAR_HIP_GEOMETRY_WORDS];
	if (cimbar_hip_geometry(ctx, geo) != CIMBAR_HIP_GEOMETRY_WORDS) { delete ing; return CIMBAR_HIP_EINVAL; }
	ing->fw = (unsigned)geo[1]; ing->fh = (unsigned)geo[2];
	ing->frame = (size_t)geo[1] * geo[2] * 3;
	ing->chunk = (unsigned)geo[5];
	ing->frame_bytes = (size_t)geo[4] * geo[5];
	// the CPUs this process may really use: hardware threads, cut down to the container's CPU quota where there is one (cgroup v2 cpu.max =
	// "<quota> <period>"; a pool larger than the quota only gets throttled -- measured on a 256-thread host with a 16-CPU quota: 32 threads
	// 2.8 k PNG frames/s, 128 threads 1.0 k)
	int hw = (int)std::thread::hardware_concurrency();
	if (hw <= 0) hw = 8;
	if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
		long long quota = 0, period = 0;
		if (std::fscanf(fq, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
			const int q = (int)((quota + period - 1) / period);
			if (q < hw) hw = q < 1 ? 1 : q;
		}
		std::fclose(fq);
	}
	ing->threads = threads > 0 ? threads : (hw > 128 ? 128 : hw);
	ing->B = batch_frames > 0 ? batch_frames : (png_mode == CIMBAR_INGEST_PNG_DEVICE ? 512 : 64);
	const int depth = cimbar_hip_pipeline_depth(ctx);
	ing->R = ring > 0 ? ring : 3;
	if (ing->R < 2) ing->R = 2;
	if (ing->R > depth) ing->R = depth;
	auto fail = [&](const char* what) { ing->err = what; cimbar_ingest_destroy(ing); return CIMBAR_HIP_ENOMEM; };
	if (hipSetDevice(ing->device) != hipSuccess) return fail("hipSetDevice");
	if (hipStreamCreateWithFlags(&ing->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail("stream");
	if (hipStreamCreateWithFlags(&ing->out_stream, hipStreamNonBlocking) != hipSuccess) return fail("stream");
	ing->slots.resize((size_t)ing->R);
	const bool dev_png = png_mode == CIMBAR_INGEST_PNG_DEVICE;
	if (dev_png) {
		// room for the batch's compressed streams: a quarter of the decoded size per frame unless the caller knows better (a frame PNG is a
		// tenth); a file that does not fit any more is skipped like one that cannot be read
		const size_t per = zbytes_per_frame ? zbytes_per_frame : ing->frame / 4;
		ing->zcap = ((size_t)ing->B * per + 4095) & ~(size_t)4095;
		if (ing->zcap > 0xFFFFF000ull) return fail("batch_frames * zbytes_per_frame exceeds 4 GiB (cimbar_hip_png_desc::pal_off is 32-bit)");
		ing->scratch_stride = cimbar_hip_png_scratch_bytes(ing->fw, ing->fh, 6);     // RGBA is the widest form a frame can arrive in
	}
	for (auto& s : ing->slots) {
		const size_t nb = (size_t)ing->B;
		if (dev_png) {
			if (hipHostMalloc((void**)&s.h_z, ing->zcap, hipHostMallocDefault) != hipSuccess) return fail("pinned PNG streams");
			if (hipMalloc((void**)&s.d_z, ing->zcap) != hipSuccess) return fail("device PNG streams");
			if (hipHostMalloc((void**)&s.h_desc, nb * sizeof(cimbar_hip_png_desc), hipHostMallocDefault) != hipSuccess) return fail("pinned descriptors");
			if (hipMalloc((void**)&s.d_desc, nb * sizeof(cimbar_hip_png_desc)) != hipSuccess) return fail("device descriptors");
			if (hipMalloc((void**)&s.d_scratch, nb * ing->scratch_stride) != hipSuccess) return fail("device scanlines");
			if (hipHostMalloc((void**)&s.h_status, nb * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return fail("pinned status");
			if (hipMalloc((void**)&s.d_status, nb * sizeof(int32_t)) != hipSuccess) return fail("device status");
			if (hipStreamCreateWithFlags(&s.png_stream, hipStreamNonBlocking) != hipSuccess) return fail("stream");
			if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess) return fail("event");
			if (hipHostMalloc((void**)&s.h_fb, (size_t)cimbar_ingest::FB * ing->frame, hipHostMallocDefault) != hipSuccess) return fail("pinned fallback frames");
			s.fb_index.assign(nb, -1);
			std::memset(s.h_desc, 0, nb * sizeof(cimbar_hip_png_desc));
		} else if (hipHostMalloc((void**)&s.h_in, nb * ing->frame, hipHostMallocDefault) != hipSuccess) return fail("pinned input");
		if (hipMalloc((void**)&s.d_in, nb * ing->frame) != hipSuccess) return fail("device input");
		if (hipHostMalloc((void**)&s.h_chunks, nb * ing->frame_bytes, hipHostMallocDefault) != hipSuccess) return fail("pinned chunks");
		if (hipMalloc((void**)&s.d_chunks, nb * ing->frame_bytes) != hipSuccess) return fail("device chunks");
		if (hipHostMalloc((void**)&s.h_masks, nb * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return fail("pinned masks");
		if (hipMalloc((void**)&s.d_masks, nb * sizeof(uint32_t)) != hipSuccess) return fail("device masks");
		if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return fail("event");
		s.valid.assign(nb, 0);
	}
	*out = ing;
	return 0;
}

void cimbar_ingest_destroy(cimbar_ingest* ing)
{
	if (!ing) return;
	(void)hipSetDevice(ing->device);
	for (auto& s : ing->slots) {
		if (s.h_in) (void)hipHostFree(s.h_in);
		if (s.d_in) (void)hipFree(s.d_in);
		if (s.h_chunks) (void)hipHostFree(s.h_chunks);
		if (s.d_chunks) (void)hipFree(s.d_chunks);
		if (s.h_masks) (void)hipHostFree(s.h_masks);
		if (s.d_masks) (void)hipFree(s.d_masks);
		if (s.done) (void)hipEventDestroy(s.done);
		if (s.h_z) (void)hipHostFree(s.h_z);
		if (s.h_fb) (void)hipHostFree(s.h_fb);
		if (s.d_z) (void)hipFree(s.d_z);
		if (s.h_desc) (void)hipHostFree(s.h_desc);
		if (s.d_desc) (void)hipFree(s.d_desc);
		if (s.d_scratch) (void)hipFree(s.d_scratch);
		if (s.h_status) (void)hipHostFree(s.h_status);
		if (s.d_status) (void)hipFree(s.d_status);
		if (s.png_stream) (void)hipStreamDestroy(s.png_stream);
		if (s.copied) (void)hipEventDestroy(s.copied);
	}
	if (ing->copy_stream) (void)hipStreamDestroy(ing->copy_stream);
	if (ing->out_stream) (void)hipStreamDestroy(ing->out_stream);
	delete ing;
}

const char* cimbar_ingest_last_error(const cimbar_ingest* ing) { return ing ? ing->err.c_str() : "null"; }

int64_t cimbar_ingest_run_files(cimbar_ingest* ing, const char* const* paths, int nfiles, int should_preprocess, int color_correction,
                                cimbar_ingest_sink_fn sink, void* user)
{
	if (!ing || !paths || nfiles < 0) return CIMBAR_HIP_EINVAL;
	if (ing->png_mode == CIMBAR_INGEST_PNG_DEVICE) {
		// the host only walks the chunks and moves the IDAT bytes; inflate + un-filter happen on the device
		auto fillz = [&](int i, cimbar_ingest::Slot& sl, int j, std::atomic<size_t>& cursor) -> bool {
			thread_local std::vector<uint8_t> file;
			thread_local PngInfo info;
			thread_local std::vector<uint8_t> idat, raw;
			cimbar_hip_png_desc& d = sl.h_desc[j];
			std::memset(&d, 0, sizeof d);                   // (zlen 0: the device refuses the slot)
			sl.fb_index[(size_t)j] = -1;
			if (!read_file(paths[i], file)) return false;
			if (png_walk(file.data(), file.size(), info) != 0) {
				// not a PNG the kernels take -- a JPEG, a 16-bit / sub-byte / interlaced PNG: this thread decodes it (what cv::imread would have
				// made of it) into one of the batch's pinned fallback frames, which joins the batch on the device
				unsigned w = 0, h = 0;
				if (image_decode(file.data(), file.size(), nullptr, 0, &w, &h, idat, raw) != 0 || w != ing->fw || h != ing->fh) return false;
				const size_t idx = cursor.fetch_add(FB_ONE) >> 48;
				if (idx >= (size_t)cimbar_ingest::FB) { ing->fb_overflow.fetch_add(1); return false; }   // more such files in one batch than it has room for: dropped, and counted as exactly that
				if (image_decode(file.data(), file.size(), sl.h_fb + idx * ing->frame, ing->frame, &w, &h, idat, raw) != 0) return false;
				sl.fb_index[(size_t)j] = (int)idx;
				return true;
			}
			if (info.w != ing->fw || info.h != ing->fh) return false;
			const size_t zal = (info.zlen + 15) & ~(size_t)15, need = zal + (info.ctype == 3 ? 768 : 0);
			const size_t off = cursor.fetch_add(need) & ZMASK;
			if (off + need > ing->zcap) return false;       // the batch's compressed bytes do not fit: see cimbar_ingest_create_ex
			uint8_t* dst = sl.h_z + off;
			for (const auto& seg : info.idat) { std::memcpy(dst, seg.first, seg.second); dst += seg.second; }
			std::memset(dst, 0, zal - info.zlen);
			if (info.ctype == 3) {
				uint8_t* pal = sl.h_z + off + zal;
				std::memset(pal, 0, 768);
				std::memcpy(pal, info.plte, (size_t)info.npal * 3);
				d.pal_off = (uint32_t)(off + zal);
			}
			d.zoff = off; d.zlen = (uint32_t)info.zlen; d.width = info.w; d.height = info.h; d.color_type = info.ctype;
			return true;
		};
		return run_pipeline(ing, nfiles, should_preprocess, color_correction, sink, user, fillz);
	}
	auto fill = [&](int i, cimbar_ingest::Slot& sl, int j, std::atomic<size_t>&) -> bool {
		thread_local std::vector<uint8_t> file, idat, raw;
		uint8_t* dst = sl.h_in + (size_t)j * ing->frame;
		if (!read_file(paths[i], file)) return false;
		unsigned w = 0, h = 0;
		if (image_decode(file.data(), file.size(), nullptr, 0, &w, &h, idat, raw) != 0) return false;
		if (w != ing->fw || h != ing->fh) return false;   // (larger, padded frames: not supported on this path)
		return image_decode(file.data(), file.size(), dst, ing->frame, &w, &h, idat, raw) == 0;
	};
	return run_pipeline(ing, nfiles, should_preprocess, color_correction, sink, user, fill);
}

int64_t cimbar_ingest_run_raw(cimbar_ingest* ing, const uint8_t* frames, int n, int should_preprocess, int color_correction,
                              cimbar_ingest_sink_fn sink, void* user)
{
	if (!ing || !frames || n < 0) return CIMBAR_HIP_EINVAL;
	if (ing->png_mode == CIMBAR_INGEST_PNG_DEVICE) { ing->err = "a device-PNG ingest has no pinned frame ring: create a host-mode one for raw frames"; return CIMBAR_HIP_EINVAL; }
	auto fill = [&](int i, cimbar_ingest::Slot& sl, int j, std::atomic<size_t>&) -> bool { std::memcpy(sl.h_in + (size_t)j * ing->frame, frames + (size_t)i * ing->frame, ing->frame); return true; };
	// frames in page-locked memory (hipHostMalloc, hipHostRegister, torch's pin_memory) are copied to the device where they lie
	hipPointerAttribute_t attr;
	const bool pinned = hipPointerGetAttributes(&attr, frames) == hipSuccess && attr.type == hipMemoryTypeHost;
	if (!pinned) (void)hipGetLastError();   // an ordinary pointer is reported as an error: clear it
	return run_pipeline(ing, n, should_preprocess, color_correction, sink, user, fill, pinned ? frames : nullptr);
}

int cimbar_ingest_png_stats(const cimbar_ingest* ing, int64_t out4[4])
{
	if (!ing || !out4) return CIMBAR_HIP_EINVAL;
	out4[0] = ing->png_files; out4[1] = ing->png_refused_host; out4[2] = ing->png_refused_device; out4[3] = ing->png_bytes;
	return 0;
}

int cimbar_ingest_timings(const cimbar_ingest* ing, double out3[3])
{
	if (!ing || !out3) return CIMBAR_HIP_EINVAL;
	out3[0] = ing->t_wall; out3[1] = ing->t_host; out3[2] = ing->t_wait;
	return 0;
}

}  // extern "C"
