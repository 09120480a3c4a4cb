// Exercises libcimbar_amd/host/Decoder.h the way the reference's callers use Decoder (cimbar.cpp:167-171, cimbar_recv_js.cpp:160-188):
// a sink with chunk_size()/write(), one frame at a time and as a batch. Built and run by tests/test_gpu_cpp_adapter.py.
#include "../../libcimbar_amd/host/Decoder.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// same STREAM concept as src/lib/encoder/escrow_buffer_writer.h:7-69
struct collecting_sink
{
	unsigned cs;
	std::vector<char> bytes;
	explicit collecting_sink(unsigned chunk) : cs(chunk) {}
	bool good() const { return true; }
	unsigned chunk_size() const { return cs; }
	long tellp() const { return (long)bytes.size(); }
	collecting_sink& write(const char* d, unsigned n) { bytes.insert(bytes.end(), d, d + n); return *this; }
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

#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv)
{
	if (argc < 5) { std::printf("usage: test_adapter frames.bin payload.bin cells.bin n\n"); return 2; }
	const int n = std::atoi(argv[4]);
	std::vector<unsigned char> frames = slurp(argv[1]), payload = slurp(argv[2]), cells = slurp(argv[3]);
	const size_t FR = 1024ull * 1024 * 3;
	CHECK(frames.size() == FR * n && payload.size() == 7500ull * n && cells.size() == 12400ull * 2);

	cimbar_amd::Decoder dec;
	CHECK(dec.good());
	CHECK(cimbar_hip_bufsize() == 7500);

	// one frame at a time, like cimbar.cpp's decode loop
	collecting_sink sink(625);
	unsigned long long total = 0;
	for (int f = 0; f < n; ++f) {
		cimbar_amd::image_view img{frames.data() + FR * f, 1024, 1024, 1024 * 3};
		unsigned got = dec.decode_fountain(img, sink);
		CHECK(got == 7500);
		total += got;
	}
	CHECK(sink.bytes.size() == payload.size());
	CHECK(std::memcmp(sink.bytes.data(), payload.data(), payload.size()) == 0);

	// as one batch
	collecting_sink sink2(625);
	CHECK(dec.decode_fountain_batch(frames.data(), n, sink2) == total);
	CHECK(sink2.bytes == sink.bytes);
	for (uint32_t m : dec.last_masks()) CHECK(m == 0xFFF);

	// Decoder::decode into a plain stream (cimbar.cpp:270-272): all 7500 bytes of every frame, tellp() as the return value
	collecting_sink plain(0);
	for (int f = 0; f < n; ++f) {
		cimbar_amd::image_view img{frames.data() + FR * f, 1024, 1024, 1024 * 3};
		CHECK(dec.decode(img, plain) == 7500u * (unsigned)(f + 1));
	}
	CHECK(plain.bytes == sink.bytes);   // clean frames: the RS outputs back to back are the fountain chunks back to back

	// chunk-size mismatch: decode, report the bytes, feed nothing (Decoder.h:180-185)
	collecting_sink wrong(600);
	cimbar_amd::image_view img0{frames.data(), 1024, 1024, 0};
	CHECK(dec.decode_fountain(img0, wrong) == 7500 && wrong.bytes.empty());

	// wrong geometry: like CimbReader::_good == false -> 0 bytes (CimbReader.cpp:119,164-167)
	cimbar_amd::image_view small{frames.data(), 512, 512, 512 * 3};
	CHECK(dec.decode_fountain(small, sink2) == 0 && dec.error_code() == CIMBAR_HIP_EDIM);

	// the stage in front (Deskewer.h:26-40, Scanner.h:148-165): paste frame 0 upright into a dark 1920x1080 capture, deskew it from the
	// anchor centres of that paste, decode the result
	{
		const int W = 1920, H = 1080, X0 = 448, Y0 = 28;
		std::vector<unsigned char> cap((size_t)W * H * 3, 0);
		for (int y = 0; y < 1024; ++y) std::memcpy(&cap[((size_t)(Y0 + y) * W + X0) * 3], frames.data() + (size_t)y * 1024 * 3, 1024 * 3);
		cimbar_amd::image_view capture{cap.data(), W, H, 0};
		struct pt { float x, y; };
		struct four { std::vector<pt> all() const { return {{478, 58}, {1442, 58}, {478, 1022}, {1442, 1022}}; } } corners;
		cimbar_amd::Deskewer de(dec);
		cimbar_amd::image bin = de.scan_preprocess(capture);
		CHECK(!bin.empty() && bin.cols == W && bin.rows == H);
		size_t on = 0;
		for (unsigned char v : bin.pixels) { CHECK(v == 0 || v == 255); on += v ? 1 : 0; }
		CHECK(on > 100000 && on < (size_t)W * H / 2);
		cimbar_amd::image fr = de.deskew(capture, corners);
		CHECK(!fr.empty() && fr.cols == 1024 && fr.rows == 1024);
		collecting_sink sink3(625);
		CHECK(dec.decode_fountain(fr, sink3) == 7500);
		CHECK(std::memcmp(sink3.bytes.data(), payload.data(), 7500) == 0);
	}

	// unsupported configuration -> !good(), decodes nothing
	cimbar_amd::Decoder no_ecc(false, true);
	CHECK(!no_ecc.good() && no_ecc.decode_fountain(img0, sink2) == 0);

	// CimbReader facade: per-cell symbol / colour bits of frame 0 against what the oracle produced for it
	cimbar_amd::CimbReader reader(img0, dec);
	CHECK(reader.num_reads() == 12400);
	unsigned count = 0;
	while (!reader.done()) {
		cimbar_amd::PositionData pos;
		unsigned sym = reader.read(pos);
		unsigned col = reader.read_color(pos);
		CHECK(sym == cells[pos.i] && col == cells[12400 + pos.i]);
		int x, y;
		cimbar_amd::CimbReader::cell_xy(pos.i, x, y);
		CHECK(pos.x == x && pos.y == y);   // clean frame: no drift
		++count;
	}
	CHECK(count == 12400);
	std::printf("OK %d frames, %llu bytes\n", n, total);
	return 0;
}
