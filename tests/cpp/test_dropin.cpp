// The drop-in proof: cimbar_amd::Decoder / Extractor / CimbReader used with the REFERENCE's own types, unmodified -- cv::Mat (through the
// oracle's cv-shim; a real OpenCV works the same way), fountain_decoder_sink and concurrent_fountain_decoder_sink (wirehair underneath) --
// the way src/exe/cimbar/cimbar.cpp:124-171,279-296 uses the reference's Decoder. Compiled in the build container against the headers
// under /root/reference (oracle/Makefile `dropin`), run on the GPU box by tests/test_gpu_cpp_adapter.py.
//
//   test_dropin frames.bin n file.bin [captures.bin w h m]          (environment: CIMBAR_MODE = 68 | 67 | 66 | 4 | 8, default 68)
// frames.bin: n frames (of the mode) of a fountain stream of file.bin (no compression). captures.bin: m camera captures of w x h.
#include "cimb_translator/Config.h"
#include "fountain/concurrent_fountain_decoder_sink.h"
#include "fountain/fountain_decoder_sink.h"

#include <opencv2/opencv.hpp>

#include "../../libcimbar_amd/host/Decoder.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

// src/exe/cimbar/cimbar.cpp:164-172 with the reference's `Decoder&` replaced by the adapter's -- the one-line type swap of INTEGRATION.md 2,
// the lambda body verbatim: a cv::UMat goes straight into decode_fountain
template <typename SINK>
std::function<int(cv::UMat, bool, int)> fountain_decode_fun(SINK& sink, cimbar_amd::Decoder& d)
{
	return [&sink, &d](cv::UMat m, bool pre, int cc) {
		return d.decode_fountain(m, sink, pre, cc);
	};
}

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

#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv)
{
	if (argc < 4) { std::printf("usage: test_dropin frames.bin n file.bin [captures.bin w h m]\n"); return 2; }
	const int n = std::atoi(argv[2]);
	std::vector<unsigned char> frames = slurp(argv[1]), file = slurp(argv[3]);
	const int MODE = std::getenv("CIMBAR_MODE") ? std::atoi(std::getenv("CIMBAR_MODE")) : 68;
	cimbar::Config::update(MODE);
	const int W = cimbar::Config::image_size_x(), H = cimbar::Config::image_size_y();
	const size_t FR = (size_t)W * H * 3;
	CHECK(frames.size() == FR * (size_t)n && !file.empty());
	const unsigned chunk = cimbar::Config::fountain_chunk_size();
	const unsigned long long FB = (unsigned long long)chunk * cimbar::Config::fountain_chunks_per_frame();
	const int NCELLS = cimbar::Config::total_cells();
	std::printf("mode %d: %dx%d, %u chunks of %u bytes\n", MODE, W, H, cimbar::Config::fountain_chunks_per_frame(), chunk);
	{
		cimbar_amd::Decoder probe(true, true, 0, MODE);
		// cimbard_get_bufsize() = fountain_chunks_per_frame() * fountain_chunk_size() of the active configuration (cimbar_recv_js.cpp:143-146)
		CHECK(probe.good() && (unsigned long long)cimbar_hip_ctx_bufsize(probe.context()) == FB && probe.fountain_chunk_size() == chunk);
		CHECK(probe.color_bits() == cimbar::Config::color_bits() && probe.symbol_bits() == cimbar::Config::symbol_bits() && probe.color_mode() == cimbar::Config::color_mode());
	}

	// 1. ./cimbar's fountain decode loop (cimbar.cpp:167-171 + :124-162 with --no-deskew): the reference's sink, our Decoder
	{
		// like ./cimbar, the sink gets an on_store callback (cimbar.cpp:279-296 passes decompress_on_store / write_on_store): without one a
		// finished stream is neither stored nor marked done (fountain_decoder_sink.h:77-96)
		std::vector<unsigned char> recovered;
		auto on_store = [&recovered](const std::string& name, const std::vector<uint8_t>& bytes) { recovered.assign(bytes.begin(), bytes.end()); return name; };
		fountain_decoder_sink sink(chunk, on_store);
		cimbar_amd::Decoder dec(true, true, 0, MODE);
		CHECK(dec.good());
		unsigned long long bytes = 0;
		int used = 0;
		for (int f = 0; f < n && sink.num_done() == 0; ++f, ++used) {
			cv::Mat img(H, W, CV_8UC3, frames.data() + FR * f);
			bytes += dec.decode_fountain(img, sink);
		}
		CHECK(bytes == FB * used);
		CHECK(sink.num_done() == 1);
		CHECK(recovered.size() == file.size() && std::memcmp(recovered.data(), file.data(), file.size()) == 0);
		CHECK(sink.is_done(FountainMetadata(9, (unsigned)file.size(), 0).id()));
		std::printf("sink: file of %zu bytes recovered from %d frames\n", file.size(), used);
	}
	// 1b. the same loop with one frame in flight (Decoder::decode_fountain_overlapped + flush: cimbar_hip_decode_frame_async / _wait underneath):
	//     a sink that records what it is given must see exactly the chunks of loop 1, in the same order, and the byte counts must add up
	{
		struct recording_sink {
			unsigned cs; std::vector<std::string> got;
			unsigned chunk_size() const { return cs; }
			bool write(const char* data, unsigned len) { got.emplace_back(data, len); return true; }
		};
		recording_sink plain{chunk, {}}, overlapped{chunk, {}};
		cimbar_amd::Decoder d1(true, true, 0, MODE), d2(true, true, 0, MODE);
		CHECK(d1.good() && d2.good());
		unsigned long long b1 = 0, b2 = 0;
		for (int f = 0; f < n; ++f) {
			cv::Mat img(H, W, CV_8UC3, frames.data() + FR * f);
			b1 += d1.decode_fountain(img, plain);
		}
		for (int f = 0; f < n; ++f) {
			// a cv::Mat of its own that dies at the end of the iteration, like cimbar.cpp:132's: the adapter must not read it after the call returns
			cv::Mat img(H, W, CV_8UC3);
			std::memcpy(img.data, frames.data() + FR * f, FR);
			const unsigned got = d2.decode_fountain_overlapped(img, overlapped);
			CHECK(f > 0 || got == 0);
			b2 += got;
			std::memset(img.data, 0x55, FR);
		}
		b2 += d2.flush(overlapped);
		CHECK(d2.flush(overlapped) == 0);
		CHECK(b1 == b2 && b1 == FB * n);
		CHECK(plain.got.size() == overlapped.got.size() && plain.got == overlapped.got);
		// and a real sink behind the overlapped loop recovers the file
		std::vector<unsigned char> recovered;
		fountain_decoder_sink sink(chunk, [&recovered](const std::string& name, const std::vector<uint8_t>& bytes) { recovered.assign(bytes.begin(), bytes.end()); return name; });
		cimbar_amd::Decoder d3(true, true, 0, MODE);
		for (int f = 0; f < n && sink.num_done() == 0; ++f) {
			cv::Mat img(H, W, CV_8UC3, frames.data() + FR * f);
			d3.decode_fountain_overlapped(img, sink);
		}
		d3.flush(sink);
		CHECK(sink.num_done() == 1 && recovered.size() == file.size() && std::memcmp(recovered.data(), file.data(), file.size()) == 0);
		std::printf("overlapped loop: %zu chunks in the order of the plain loop, file recovered\n", overlapped.got.size());
	}
	// 2. the same through concurrent_fountain_decoder_sink (what cimbar_recv feeds from its worker threads)
	{
		std::vector<unsigned char> recovered;
		concurrent_fountain_decoder_sink sink(chunk, [&recovered](const std::string& name, const std::vector<uint8_t>& bytes) { recovered.assign(bytes.begin(), bytes.end()); return name; });
		cimbar_amd::Decoder dec(true, true, 0, MODE);
		for (int f = 0; f < n && sink.num_done() == 0; ++f) {
			cv::Mat img(H, W, CV_8UC3, frames.data() + FR * f);
			dec.decode_fountain(img, sink);
			sink.process();
		}
		CHECK(sink.num_done() == 1);
		CHECK(recovered.size() == file.size() && std::memcmp(recovered.data(), file.data(), file.size()) == 0);
		std::printf("concurrent sink: done\n");
	}
	// 3. batch entry point feeding the same sink type
	{
		int stored = 0;
		fountain_decoder_sink sink(chunk, [&stored](const std::string& name, const std::vector<uint8_t>&) { ++stored; return name; });
		cimbar_amd::Decoder dec(true, true, 0, MODE);
		unsigned long long bytes = dec.decode_fountain_batch(frames.data(), n, sink);     // chunks after completion are ignored by the sink (:146-148)
		CHECK(bytes == FB * n);
		CHECK(sink.num_done() == 1 && stored == 1);
	}
	// 4. CimbReader in the reference's constructor shape (CimbReader.h:16-17)
	{
		cimbar_amd::Decoder dec(true, true, 0, MODE);
		cimbar_amd::CimbDecoder cd(dec, cimbar::Config::symbol_bits(), cimbar::Config::color_bits());
		CHECK(cd.good());
		cv::Mat img(H, W, CV_8UC3, frames.data());
		cimbar_amd::CimbReader reader(img, cd, cimbar::Config::color_mode());
		CHECK((int)reader.num_reads() == NCELLS);
		unsigned count = 0;
		while (!reader.done()) { cimbar_amd::PositionData pos; unsigned bits = reader.read(pos); CHECK(bits < 16 && reader.read_color(pos) < (1u << cimbar::Config::color_bits())); ++count; }
		CHECK((int)count == NCELLS);
		// ... and from a cv::UMat (CimbReader.h:17)
		cv::UMat um = cv::getUMat(img, cv::ACCESS_RW);
		cimbar_amd::CimbReader ureader(um, cd, cimbar::Config::color_mode());
		CHECK((int)ureader.num_reads() == NCELLS && !ureader.done());
	}
	// 4b. cimbar.cpp's decode loop as it is written there (:124-171): cv::UMat all the way, std::function<int(cv::UMat,bool,int)>, the
	//     reference's sink -- INTEGRATION.md 2's type swap compiles and runs verbatim
	{
		std::vector<unsigned char> recovered;
		fountain_decoder_sink sink(chunk, [&recovered](const std::string& name, const std::vector<uint8_t>& bytes) { recovered.assign(bytes.begin(), bytes.end()); return name; });
		cimbar_amd::Decoder d(true, true, 0, MODE);
		std::function<int(cv::UMat, bool, int)> decodefun = fountain_decode_fun(sink, d);
		int err = 0;
		for (int f = 0; f < n && sink.num_done() == 0; ++f) {
			cv::UMat img = cv::getUMat(cv::Mat(H, W, CV_8UC3, frames.data() + FR * f), cv::ACCESS_RW);   // cimbar.cpp:132: cv::imread(inf).getUMat(cv::ACCESS_RW)
			int bytes = decodefun(img, false, 2);
			if (!bytes) err |= 4;
		}
		CHECK(err == 0 && sink.num_done() == 1);
		CHECK(recovered.size() == file.size() && std::memcmp(recovered.data(), file.data(), file.size()) == 0);
		std::printf("cv::UMat loop: done\n");
	}
	// 5. Extractor on cv::Mat captures, then the decode (cimbar.cpp:139-158)
	if (argc >= 8) {
		std::vector<unsigned char> caps = slurp(argv[4]);
		const int w = std::atoi(argv[5]), h = std::atoi(argv[6]), m = std::atoi(argv[7]);
		CHECK(caps.size() == (size_t)w * h * 3 * m);
		cimbar_amd::Decoder dec(true, true, 0, MODE);
		cimbar_amd::Extractor ext(dec);
		fountain_decoder_sink sink(chunk, [](const std::string& name, const std::vector<uint8_t>&) { return name; });
		int extracted = 0;
		for (int k = 0; k < m; ++k) {
			cv::Mat img(h, w, CV_8UC3, caps.data() + (size_t)w * h * 3 * k);
			cv::Mat out;
			int res = ext.extract(img, out);
			if (!res) continue;
			++extracted;
			CHECK(out.rows == H && out.cols == W);
			bool pre = res == cimbar_amd::Extractor::NEEDS_SHARPEN;
			const unsigned got = dec.decode_fountain(out, sink, pre);
			std::printf("capture %d: extract %d, decoded %u bytes\n", k, res, got);
			// cimbar.cpp:132-155 as written: a cv::UMat, extracted IN PLACE (`ext.extract(img, img)`), then decoded
			cv::UMat u = cv::getUMat(img.clone(), cv::ACCESS_RW);
			int res2 = ext.extract(u, u);
			CHECK(res2 == res && u.rows == H && u.cols == W);
			CHECK(std::memcmp(u.getMat(cv::ACCESS_READ).data, out.data, FR) == 0);
			fountain_decoder_sink sink2(chunk, [](const std::string& name, const std::vector<uint8_t>&) { return name; });
			CHECK(dec.decode_fountain(u, sink2, pre) == got);
		}
		CHECK(extracted >= 1);
	}
	std::printf("OK\n");
	return 0;
}
