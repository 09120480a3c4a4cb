// libcimbar_amd/host/Decoder.h in EVERY mode cimbar_hip_create accepts (68, 67, 66, 4, 8 = every mode of Config::temp_conf, Config.h:19-44), the
// way the reference's callers size and drive things: chunk space by cimbard_get_bufsize() of the active configuration
// (cimbar_recv_js.cpp:143-146), CimbDecoder(Config::symbol_bits(), Config::color_bits()) + CimbReader(img, decoder, Config::color_mode()),
// DecoderPlus::load_ccm / save_ccm (DecoderPlus.h:32-58). Built twice by tests/test_gpu_cpp_adapter.py: plain, and with -fsanitize=address
// (mode 8's frame is 8750 bytes -- more than mode B's 7500, the size every fixed buffer of round 2 had).
//
//   test_adapter_modes mode frames.bin payload.bin cells.bin n tmpdir
// frames.bin: n clean frames of the mode; payload.bin: the bytes they carry; cells.bin: symbol bits then colour bits of frame 0, per linear cell.
#include "../../libcimbar_amd/host/Decoder.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct collecting_sink   // the STREAM concept of src/lib/encoder/escrow_buffer_writer.h:7-69
{
	unsigned cs;
	std::vector<char> bytes;
	explicit collecting_sink(unsigned chunk) : cs(chunk) {}
	bool good() const { return true; }
	unsigned chunk_size() const { return cs; }
	long tellp() const { return (long)bytes.size(); }
	collecting_sink& write(const char* d, unsigned n) { bytes.insert(bytes.end(), d, d + n); return *this; }
};

// something shaped like cv::UMat: no data / step of its own, only getMat(access) -- with an enum argument like OpenCV 4's
enum AccessFlag { ACCESS_READ = 1 << 24, ACCESS_WRITE = 1 << 25, ACCESS_RW = 3 << 24 };
struct device_image
{
	cimbar_amd::image host;
	int last_access = 0;
	int cols = 0, rows = 0;
	cimbar_amd::image getMat(AccessFlag a) const { const_cast<device_image*>(this)->last_access = (int)a; return host; }
	int type() const { return host.type(); }
};

static std::vector<unsigned char> slurp(const char* path)
{
	std::vector<unsigned char> v;
	FILE* f = std::fopen(path, "rb");
	if (!f) return v;
	std::fseek(f, 0, SEEK_END);
	long n = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	v.resize((size_t)n);
	if (std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
	std::fclose(f);
	return v;
}

#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL mode %d line %d: %s\n", mode, __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv)
{
	int mode = 0;
	if (argc < 7) { std::printf("usage: test_adapter_modes mode frames.bin payload.bin cells.bin n tmpdir\n"); return 2; }
	mode = std::atoi(argv[1]);
	const int n = std::atoi(argv[5]);
	const std::string tmp = argv[6];
	std::vector<unsigned char> frames = slurp(argv[2]), payload = slurp(argv[3]), cells = slurp(argv[4]);

	struct expect { int w, h, chunks, chunk, cells, color_bits, color_mode; };
	const expect E = mode == 67 ? expect{1024, 720, 12, 429, 8592, 2, 1} : mode == 66 ? expect{736, 637, 6, 540, 5376, 2, 1}
	               : mode == 4 ? expect{1024, 1024, 10, 750, 12400, 2, 0} : mode == 8 ? expect{1024, 1024, 10, 875, 12400, 3, 0}
	                           : expect{1024, 1024, 12, 625, 12400, 2, 1};
	const unsigned FB = (unsigned)E.chunks * E.chunk;
	const size_t FR = (size_t)E.w * E.h * 3;
	CHECK(frames.size() == FR * n && payload.size() == (size_t)FB * n && cells.size() == (size_t)E.cells * 2);

	cimbar_amd::Decoder dec(true, true, 0, mode);
	CHECK(dec.good());
	// cimbard_get_bufsize() of the active configuration; the context-free call keeps answering for the default configuration
	CHECK(cimbar_hip_ctx_bufsize(dec.context()) == (int)FB && dec.frame_bytes() == FB && cimbar_hip_bufsize() == 7500);
	CHECK(FB <= CIMBAR_HIP_MAX_FRAME_BYTES);
	CHECK((int)dec.image_size_x() == E.w && (int)dec.image_size_y() == E.h && (int)dec.fountain_chunks_per_frame() == E.chunks &&
	      (int)dec.fountain_chunk_size() == E.chunk && (int)dec.total_cells() == E.cells);
	CHECK((int)dec.color_bits() == E.color_bits && (int)dec.color_mode() == E.color_mode && dec.symbol_bits() == 4);

	// the C call with a buffer of exactly cimbar_hip_ctx_bufsize bytes between two canaries
	{
		std::vector<unsigned char> buf(FB + 64, 0xA5);
		uint32_t mask = 0;
		int r = cimbar_hip_decode_frame(dec.context(), frames.data(), (unsigned)E.w, (unsigned)E.h, 0, 0, 2, buf.data() + 32, &mask);
		CHECK(r == (int)FB && mask == (1u << E.chunks) - 1u);
		for (int k = 0; k < 32; ++k) CHECK(buf[k] == 0xA5 && buf[32 + FB + k] == 0xA5);
		CHECK(std::memcmp(buf.data() + 32, payload.data(), FB) == 0);
		// a too-small image: zeros, full count (CimbReader.cpp:119) -- the memset path writes the context's frame size too
		std::fill(buf.begin(), buf.end(), 0xA5);
		r = cimbar_hip_decode_frame(dec.context(), frames.data(), 64, 64, 0, 0, 2, buf.data() + 32, &mask);
		CHECK(r == (int)FB);
		for (int k = 0; k < 32; ++k) CHECK(buf[k] == 0xA5 && buf[32 + FB + k] == 0xA5);
	}

	// one frame at a time, like cimbar.cpp's decode loop
	collecting_sink sink((unsigned)E.chunk);
	unsigned long long total = 0;
	for (int f = 0; f < n; ++f) {
		cimbar_amd::image_view img{frames.data() + FR * f, E.w, E.h, (size_t)E.w * 3};
		unsigned got = dec.decode_fountain(img, sink);
		CHECK(got == FB);
		total += got;
	}
	CHECK(sink.bytes.size() == payload.size() && std::memcmp(sink.bytes.data(), payload.data(), payload.size()) == 0);

	// as one batch
	collecting_sink sink2((unsigned)E.chunk);
	CHECK(dec.decode_fountain_batch(frames.data(), n, sink2) == total && sink2.bytes == sink.bytes);
	for (uint32_t m : dec.last_masks()) CHECK(m == (1u << E.chunks) - 1u);

	// Decoder::decode into a plain stream (cimbar.cpp:270-272)
	collecting_sink plain(0);
	for (int f = 0; f < n; ++f) {
		cimbar_amd::image_view img{frames.data() + FR * f, E.w, E.h, (size_t)E.w * 3};
		CHECK(dec.decode(img, plain) == FB * (unsigned)(f + 1));
	}
	CHECK(plain.bytes == sink.bytes);

	// the same through something shaped like cv::UMat (cimbar.cpp:132,167-171): only getMat(access) leads to the pixels
	{
		device_image u;
		u.host = cimbar_amd::image(E.w, E.h, 3);
		std::memcpy(u.host.data, frames.data(), FR);
		u.cols = E.w; u.rows = E.h;
		collecting_sink s3((unsigned)E.chunk), p3(0);
		CHECK(dec.decode_fountain(u, s3) == FB && u.last_access == ACCESS_READ);
		CHECK(std::memcmp(s3.bytes.data(), payload.data(), FB) == 0);
		CHECK(dec.decode(u, p3) == FB && std::memcmp(p3.bytes.data(), payload.data(), FB) == 0);
		cimbar_amd::CimbDecoder cd(dec, dec.symbol_bits(), dec.color_bits());
		cimbar_amd::CimbReader r(u, cd, dec.color_mode());
		CHECK(!r.done() && (int)r.num_reads() == E.cells);
	}

	// CimbReader in the reference's constructor shape, with the mode's bit counts (CimbReader.h:16-17, Decoder.h:40-45)
	{
		cimbar_amd::image_view img0{frames.data(), E.w, E.h, 0};
		cimbar_amd::CimbDecoder cd(dec, 4, (unsigned)E.color_bits);
		CHECK(cd.good());
		cimbar_amd::CimbReader reader(img0, cd, (unsigned)E.color_mode);
		CHECK((int)reader.num_reads() == E.cells);
		unsigned count = 0;
		while (!reader.done()) {
			cimbar_amd::PositionData pos;
			unsigned sym = reader.read(pos);
			unsigned col = reader.read_color(pos);
			CHECK(sym == cells[pos.i] && col == cells[(size_t)E.cells + pos.i]);
			++count;
		}
		CHECK((int)count == E.cells);
		// the wrong bit counts / colour mode for this configuration: not good, reads nothing
		cimbar_amd::CimbDecoder wrong(dec, 4, (unsigned)E.color_bits == 2 ? 3u : 2u);
		CHECK(!wrong.good());
		cimbar_amd::CimbReader none(img0, wrong, (unsigned)E.color_mode);
		CHECK(none.done());
		cimbar_amd::CimbReader none2(img0, cd, (unsigned)E.color_mode ^ 1u);
		CHECK(none2.done());
	}

	// DecoderPlus::load_ccm / save_ccm (DecoderPlus.h:32-58): nine floats; save only while a matrix is active
	{
		const std::string path = tmp + "/ccm.bin", back = tmp + "/ccm_back.bin", shortf = tmp + "/ccm_short.bin";
		cimbar_hip_reset_ccm(dec.context());
		CHECK(!dec.save_ccm(back));                                     // `not get_ccm().active()` -> false
		CHECK(!dec.load_ccm(tmp + "/does_not_exist.bin"));
		const float m[9] = {1.25f, -0.125f, 0.0f, 0.03125f, 0.875f, 0.0625f, -0.25f, 0.5f, 1.5f};
		FILE* f = std::fopen(path.c_str(), "wb"); CHECK(f); std::fwrite(m, 1, sizeof m, f); std::fclose(f);
		f = std::fopen(shortf.c_str(), "wb"); CHECK(f); std::fwrite(m, 1, 35, f); std::fclose(f);
		CHECK(!dec.load_ccm(shortf));                                   // `data.size() < 3*3*4`
		CHECK(dec.load_ccm(path));
		float got[9];
		CHECK(cimbar_hip_get_ccm(dec.context(), got) == 1 && std::memcmp(got, m, sizeof m) == 0);
		CHECK(dec.save_ccm(back));
		std::vector<unsigned char> raw = slurp(back.c_str());
		CHECK(raw.size() == sizeof m && std::memcmp(raw.data(), m, sizeof m) == 0);
		// the loaded matrix is in force for a --no-fountain decode (cimbar.cpp:263-272): colour bits change under a channel-swapping matrix ...
		const float swap[9] = {0, 1, 0, 1, 0, 0, 0, 0, 1};
		CHECK(dec.update_color_correction(swap));
		cimbar_amd::image_view img0{frames.data(), E.w, E.h, 0};
		cimbar_amd::CimbReader swapped(img0, dec, false, 0);
		unsigned differ = 0;
		while (!swapped.done()) { cimbar_amd::PositionData pos; swapped.read(pos); differ += swapped.read_color(pos) != cells[(size_t)E.cells + pos.i]; }
		CHECK(differ > (unsigned)E.cells / 8);
		// ... and the identity matrix gives the plain classification back
		const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		CHECK(dec.update_color_correction(ident));
		cimbar_amd::CimbReader same(img0, dec, false, 0);
		differ = 0;
		while (!same.done()) { cimbar_amd::PositionData pos; same.read(pos); differ += same.read_color(pos) != cells[(size_t)E.cells + pos.i]; }
		CHECK(differ == 0);
		cimbar_hip_reset_ccm(dec.context());
	}
	std::printf("OK mode %d: %d frames, %llu bytes, bufsize %u\n", mode, n, total, FB);
	return 0;
}
