// The drop-in proof: cimbar_amd::Decoder / Extractor / CimbReader used with the REFERENCE's own types, unmodified -- cv::Mat (through the
// oracle's cv-shim; a real OpenCV works the same way), fountain_decoder_sink and concurrent_fountain_decoder_sink (wirehair underneath) --
// the way src/exe/cimbar/cimbar.cpp:124-171,279-296 uses the reference's Decoder. Compiled in the build container against the headers
// under /root/reference (oracle/Makefile `dropin`), run on the GPU box by tests/test_gpu_cpp_adapter.py.
//
//   test_dropin frames.bin n file.bin [captures.bin w h m]
// frames.bin: n mode-B frames of a fountain stream of file.bin (no compression). captures.bin: m camera captures of w x h.
#include "cimb_translator/Config.h"
#include "fountain/concurrent_fountain_decoder_sink.h"
#include "fountain/fountain_decoder_sink.h"

#include <opencv2/opencv.hpp>

#include "../../libcimbar_amd/host/Decoder.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
	const size_t FR = 1024ull * 1024 * 3;
	CHECK(frames.size() == FR * (size_t)n && !file.empty());
	cimbar::Config::update(68);
	const unsigned chunk = cimbar::Config::fountain_chunk_size();
	CHECK(chunk == 625);

	// 1. ./cimbar's fountain decode loop (cimbar.cpp:167-171 + :124-162 with --no-deskew): the reference's sink, our Decoder
	{
		// like ./cimbar, the sink gets an on_store callback (cimbar.cpp:279-296 passes decompress_on_store / write_on_store): without one a
		// finished stream is neither stored nor marked done (fountain_decoder_sink.h:77-96)
		std::vector<unsigned char> recovered;
		auto on_store = [&recovered](const std::string& name, const std::vector<uint8_t>& bytes) { recovered.assign(bytes.begin(), bytes.end()); return name; };
		fountain_decoder_sink sink(chunk, on_store);
		cimbar_amd::Decoder dec;
		CHECK(dec.good());
		unsigned long long bytes = 0;
		int used = 0;
		for (int f = 0; f < n && sink.num_done() == 0; ++f, ++used) {
			cv::Mat img(1024, 1024, CV_8UC3, frames.data() + FR * f);
			bytes += dec.decode_fountain(img, sink);
		}
		CHECK(bytes == 7500ull * used);
		CHECK(sink.num_done() == 1);
		CHECK(recovered.size() == file.size() && std::memcmp(recovered.data(), file.data(), file.size()) == 0);
		CHECK(sink.is_done(FountainMetadata(9, (unsigned)file.size(), 0).id()));
		std::printf("sink: file of %zu bytes recovered from %d frames\n", file.size(), used);
	}
	// 2. the same through concurrent_fountain_decoder_sink (what cimbar_recv feeds from its worker threads)
	{
		std::vector<unsigned char> recovered;
		concurrent_fountain_decoder_sink sink(chunk, [&recovered](const std::string& name, const std::vector<uint8_t>& bytes) { recovered.assign(bytes.begin(), bytes.end()); return name; });
		cimbar_amd::Decoder dec;
		for (int f = 0; f < n && sink.num_done() == 0; ++f) {
			cv::Mat img(1024, 1024, CV_8UC3, frames.data() + FR * f);
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
		cimbar_amd::Decoder dec;
		unsigned long long bytes = dec.decode_fountain_batch(frames.data(), n, sink);     // chunks after completion are ignored by the sink (:146-148)
		CHECK(bytes == 7500ull * n);
		CHECK(sink.num_done() == 1 && stored == 1);
	}
	// 4. CimbReader in the reference's constructor shape (CimbReader.h:16-17)
	{
		cimbar_amd::Decoder dec;
		cimbar_amd::CimbDecoder cd(dec, cimbar::Config::symbol_bits(), cimbar::Config::color_bits());
		cv::Mat img(1024, 1024, CV_8UC3, frames.data());
		cimbar_amd::CimbReader reader(img, cd, cimbar::Config::color_mode());
		CHECK(reader.num_reads() == 12400);
		unsigned count = 0;
		while (!reader.done()) { cimbar_amd::PositionData pos; unsigned bits = reader.read(pos); CHECK(bits < 16 && reader.read_color(pos) < 4); ++count; }
		CHECK(count == 12400);
	}
	// 5. Extractor on cv::Mat captures, then the decode (cimbar.cpp:139-158)
	if (argc >= 8) {
		std::vector<unsigned char> caps = slurp(argv[4]);
		const int w = std::atoi(argv[5]), h = std::atoi(argv[6]), m = std::atoi(argv[7]);
		CHECK(caps.size() == (size_t)w * h * 3 * m);
		cimbar_amd::Decoder dec;
		cimbar_amd::Extractor ext(dec);
		fountain_decoder_sink sink(chunk, [](const std::string& name, const std::vector<uint8_t>&) { return name; });
		int extracted = 0;
		for (int k = 0; k < m; ++k) {
			cv::Mat img(h, w, CV_8UC3, caps.data() + (size_t)w * h * 3 * k);
			cv::Mat out;
			int res = ext.extract(img, out);
			if (!res) continue;
			++extracted;
			CHECK(out.rows == 1024 && out.cols == 1024);
			bool pre = res == cimbar_amd::Extractor::NEEDS_SHARPEN;
			std::printf("capture %d: extract %d, decoded %u bytes\n", k, res, dec.decode_fountain(out, sink, pre));
		}
		CHECK(extracted >= 1);
	}
	std::printf("OK\n");
	return 0;
}
