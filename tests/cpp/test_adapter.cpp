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

	// an image smaller than the frame: CimbReader::_good == false (CimbReader.cpp:119) -> no cell is read, and the reference's Reed-Solomon pass
	// then delivers its all-zero buffers as twelve chunks of zeros with the full byte count; same here
	cimbar_amd::image_view small{frames.data(), 512, 512, 512 * 3};
	collecting_sink zeros(625);
	CHECK(dec.decode_fountain(small, zeros) == 7500 && zeros.bytes.size() == 7500);
	for (unsigned char b : zeros.bytes) CHECK(b == 0);

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
		reader.cell_xy(pos.i, x, y);
		CHECK(pos.x == x && pos.y == y);   // clean frame: no drift
		++count;
	}
	CHECK(count == 12400);

	// mode 67 ("Bm", 1024x720, 12 x 429 bytes) through the same classes: argv[5] = frames, argv[6] = payload, argv[7] = a 1080p capture of frame 0
	if (argc >= 8) {
		std::vector<unsigned char> f67 = slurp(argv[5]), p67 = slurp(argv[6]), cap = slurp(argv[7]);
		cimbar_amd::Decoder mini(true, true, 0, 67);
		CHECK(mini.good() && mini.image_size_x() == 1024 && mini.image_size_y() == 720 && mini.fountain_chunk_size() == 429 && mini.total_cells() == 8592);
		const size_t FM = 1024ull * 720 * 3;
		CHECK(f67.size() == FM * n && p67.size() == 5148ull * n && cap.size() == 1920ull * 1080 * 3);
		collecting_sink s67(429), wrong(625);
		for (int f = 0; f < n; ++f) {
			cimbar_amd::image_view img{f67.data() + FM * f, 1024, 720, 1024 * 3};
			CHECK(mini.decode_fountain(img, s67) == 5148);
			CHECK(dec.decode_fountain(img, wrong) == 7500 && wrong.bytes.size() == 7500ull * (f + 1) && wrong.bytes[7500ull * f + 17] == 0);   // "too small" for mode B: zero chunks, like the reference
		}
		CHECK(s67.bytes.size() == p67.size() && std::memcmp(s67.bytes.data(), p67.data(), p67.size()) == 0);
		collecting_sink b67(429);
		CHECK(mini.decode_fountain_batch(f67.data(), n, b67) == 5148ull * n && b67.bytes == s67.bytes);
		cimbar_amd::image capture(1920, 1080, 3);          // Extractor::extract(const MAT&, MAT&): the same image type in and out
		std::memcpy(capture.data, cap.data(), cap.size());
		cimbar_amd::Extractor ext(mini);
		cimbar_amd::image deskewed;
		CHECK(ext.extract(capture, deskewed) != cimbar_amd::Extractor::FAILURE && deskewed.cols == 1024 && deskewed.rows == 720);
		collecting_sink c67(429);
		CHECK(mini.decode_fountain(deskewed, c67) == 5148 && std::memcmp(c67.bytes.data(), p67.data(), 5148) == 0);
		cimbar_amd::CimbReader r67(cimbar_amd::image_view{f67.data(), 1024, 720, 1024 * 3}, mini);
		CHECK(r67.num_reads() == 8592);
		cimbar_amd::PositionData p0;
		r67.read(p0);
		CHECK(p0.x == 9 + 54 && p0.y == 9);   // CellPositions: first cell right of the top-left marker, cell_offset 9
	}
	std::printf("OK %d frames, %llu bytes\n", n, total);
	return 0;
}
