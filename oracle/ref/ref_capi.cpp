// oracle/_ref/libcimbar_ref.so -- a flat C API around the REFERENCE's own classes, compiled from the sources
// where they lie under /root/reference (see oracle/Makefile) against oracle/cvshim (OpenCV stand-in).
//
// TEST INFRASTRUCTURE ONLY: used to pin oracle/cimbar_oracle.c, to manufacture golden vectors / input frames,
// and (optionally) as the "reference" CPU baseline in bench.py. Never linked or loaded by libcimbar_amd.
//
// Nothing here re-implements reference logic except the 6-line body of Decoder::decode_fountain
// (src/lib/encoder/Decoder.h:171-189), repeated in RefDecoder::decode_fountain_masked so that the per-chunk
// good/bad outcome (which aligned_stream only reports through its callback) can be observed.
#include "cimb_translator/CimbDecoder.h"
#include "cimb_translator/CimbReader.h"
#include "cimb_translator/CimbWriter.h"
#include "cimb_translator/Common.h"
#include "cimb_translator/Config.h"
#include "cimb_translator/Interleave.h"
#include "encoder/Decoder.h"
#include "encoder/Encoder.h"
#include "encoder/ReedSolomon.h"
#include "encoder/escrow_buffer_writer.h"
#include "extractor/Corners.h"
#include "extractor/Deskewer.h"
#include "extractor/Extractor.h"
#include "extractor/Scanner.h"
#include "fountain/fountain_decoder_sink.h"
#include <chrono>
#include "fountain/fountain_encoder_stream.h"
#include "compression/zstd_decompressor.h"

#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <vector>

namespace {

// collects aligned_stream's good-chunk writes (same STREAM concept as fountain_decoder_sink / escrow_buffer_writer)
struct chunk_collector
{
	unsigned _chunk;
	std::vector<char> bytes;
	long count = 0;
	explicit chunk_collector(unsigned chunk) : _chunk(chunk) {}
	bool good() const { return true; }
	unsigned chunk_size() const { return _chunk; }
	long tellp() const { return count; }
	chunk_collector& write(const char* d, unsigned n) { bytes.insert(bytes.end(), d, d + n); count += n; return *this; }
};

struct TestableCimbDecoder : CimbDecoder
{
	using CimbDecoder::CimbDecoder;
	void reset_ccm() { internal_ccm() = color_correction(); }
	const std::vector<uint64_t>& hashes() const { return _tileHashes; }
};

struct ExposedReader : CimbReader
{
	using CimbReader::CimbReader;
	const std::vector<char>& bitplane() const { return _grayscale.buffer(); }
};

// Sits where do_decode expects its STREAM (between reed_solomon_stream and the reference's aligned_stream) and only
// counts RS blocks, so that a chunk delivered by aligned_stream can be attributed to its slot (block group / 5).
struct block_counting_tee
{
	aligned_stream<chunk_collector>& inner;
	unsigned blocks = 0;
	explicit block_counting_tee(aligned_stream<chunk_collector>& a) : inner(a) {}
	bool good() const { return inner.good(); }
	long tellp() const { return inner.tellp(); }
	block_counting_tee& write(const char* d, unsigned n) { blocks += 1; inner.write(d, n); return *this; }
};
inline block_counting_tee& operator<<(block_counting_tee& s, const ReedSolomon::BadChunk& chunk)
{
	s.blocks += 1;
	s.inner.mark_bad_chunk(chunk.size);   // == operator<<(aligned_stream&, BadChunk), reed_solomon_stream.h:109-114
	return s;
}

struct RefDecoder : Decoder
{
	using Decoder::Decoder;

	// Decoder::decode_fountain (Decoder.h:171-189), same objects (CimbReader, aligned_stream bound to update_metadata,
	// do_decode); the tee records which group of blocks each delivered chunk came from.
	unsigned decode_fountain_masked(const cv::Mat& img, chunk_collector& out, std::vector<unsigned>& slots, bool pre, int cc)
	{
		CimbReader reader(img, _decoder, cimbar::Config::color_mode(), pre, cc);
		unsigned chunk_size = cimbar::Config::fountain_chunk_size();
		const unsigned blocks_per_chunk = chunk_size / (cimbar::Config::ecc_block_size() - cimbar::Config::ecc_bytes());   // 5 (mode B) | 3 (Bm)
		block_counting_tee* teep = nullptr;
		auto on_flush = [&](char* buf, size_t len) {
			reader.update_metadata(buf, len, chunk_size);
			if (buf != nullptr && len > 0) slots.push_back((teep->blocks - 1) / blocks_per_chunk);
		};
		aligned_stream<chunk_collector> aligner(out, out.chunk_size(), 0, on_flush);
		block_counting_tee tee(aligner);
		teep = &tee;
		return do_decode(reader, tee);
	}
};

void reset_thread_ccm()
{
	TestableCimbDecoder d(cimbar::Config::symbol_bits(), cimbar::Config::color_bits(), cimbar::Config::dark(), 0xFF);
	d.reset_ccm();
}

std::shared_ptr<fountain_decoder_sink> g_sink;
}

extern "C" {

int ref_configure(int mode_val)
{
	cimbar::Config::update(mode_val);
	return 0;
}

// the 16 tile hashes as CimbDecoder computes them at construction (CimbDecoder.cpp:87-99)
int ref_tile_hashes(uint64_t* out16)
{
	TestableCimbDecoder d(cimbar::Config::symbol_bits(), cimbar::Config::color_bits(), cimbar::Config::dark(), 0xFF);
	for (unsigned i = 0; i < d.hashes().size() && i < 16; ++i) out16[i] = d.hashes()[i];
	return (int)d.hashes().size();
}

// tile `color*16+symbol` rendered exactly as the encoder pastes it (CimbEncoder.cpp:20-25 -> Common.cpp:150-171); 8*8*3 bytes
int ref_tile_rgb(unsigned symbol, unsigned color, unsigned color_mode, uint8_t* out192)
{
	cv::Mat t = cimbar::getTile(cimbar::Config::symbol_bits(), symbol, cimbar::Config::dark(), 1u << cimbar::Config::color_bits(), color, color_mode);
	if (t.rows != 8 || t.cols != 8 || t.channels() != 3) return -1;
	for (int y = 0; y < 8; ++y) std::memcpy(out192 + y * 24, t.ptr<uchar>(y), 24);
	return 0;
}

// an empty frame: background + anchors + guides, no cells (CimbWriter.cpp:39-77)
int ref_template_frame(uint8_t* out_rgb)
{
	CimbWriter w(cimbar::Config::symbol_bits(), cimbar::Config::color_bits(), cimbar::Config::dark(), cimbar::Config::color_mode());
	cv::Mat img = w.image();
	for (int y = 0; y < img.rows; ++y) std::memcpy(out_rgb + (size_t)y * img.cols * 3, img.ptr<uchar>(y), (size_t)img.cols * 3);
	return img.rows * img.cols * 3;
}

// Encoder::encode_next (Encoder.h:69-129) over a raw byte stream: `n` payload bytes (<= 7500 per frame) -> one RGB frame
int ref_encode_raw(const uint8_t* payload, unsigned n, uint8_t* out_rgb)
{
	std::stringstream ss(std::string((const char*)payload, n));
	Encoder enc;
	auto frame = enc.encode_next(ss);
	if (!frame) return -1;
	for (int y = 0; y < frame->rows; ++y) std::memcpy(out_rgb + (size_t)y * frame->cols * 3, frame->ptr<uchar>(y), (size_t)frame->cols * 3);
	return frame->rows * frame->cols * 3;
}

// fountain-encode `data` (compression off) and render `nframes` consecutive frames (Encoder.h:167-190, :69-129;
// EncoderPlus.h:45-98 minus the will_it_scan filter, which only matters when a Scanner is in front of the decoder)
int ref_encode_fountain(const uint8_t* data, unsigned size, int encode_id, unsigned first_frame, unsigned nframes, uint8_t* out_rgb)
{
	std::stringstream ss(std::string((const char*)data, size));
	Encoder enc;
	enc.set_encode_id((uint8_t)encode_id);
	fountain_encoder_stream::ptr fes = enc.create_fountain_encoder(ss, "", 0);
	if (!fes) return -1;
	size_t fsz = (size_t)cimbar::Config::image_size_x() * cimbar::Config::image_size_y() * 3;
	for (unsigned f = 0; f < first_frame + nframes; ++f)
	{
		auto frame = enc.encode_next(*fes);
		if (!frame) return (int)f;
		if (f < first_frame) continue;
		uint8_t* dst = out_rgb + (size_t)(f - first_frame) * fsz;
		for (int y = 0; y < frame->rows; ++y) std::memcpy(dst + (size_t)y * frame->cols * 3, frame->ptr<uchar>(y), (size_t)frame->cols * 3);
	}
	return (int)nframes;
}

// BASELINE configs[0] as `./cimbar --encode` makes it (cimbar.cpp:106-121 -> EncoderPlus.h:45-98): zstd at `compression` with the
// file's basename in a skippable header frame, encode id as given (the CLI uses 109), minus the will_it_scan filter
int ref_encode_fountain_z(const uint8_t* data, unsigned size, int encode_id, int compression, const char* basename, unsigned first_frame,
                          unsigned nframes, uint8_t* out_rgb)
{
	std::stringstream ss(std::string((const char*)data, size));
	Encoder enc;
	enc.set_encode_id((uint8_t)encode_id);
	fountain_encoder_stream::ptr fes = enc.create_fountain_encoder(ss, basename ? basename : "", compression);
	if (!fes) return -1;
	size_t fsz = (size_t)cimbar::Config::image_size_x() * cimbar::Config::image_size_y() * 3;
	for (unsigned f = 0; f < first_frame + nframes; ++f)
	{
		auto frame = enc.encode_next(*fes);
		if (!frame) return (int)f;
		if (f < first_frame) continue;
		uint8_t* dst = out_rgb + (size_t)(f - first_frame) * fsz;
		for (int y = 0; y < frame->rows; ++y) std::memcpy(dst + (size_t)y * frame->cols * 3, frame->ptr<uchar>(y), (size_t)frame->cols * 3);
	}
	return (int)nframes;
}

// what the decode side of the CLI does with a recovered fountain file (cimbar.cpp: decompress_on_store -> zstd_decompressor)
int ref_zstd_decompress(const uint8_t* in, unsigned n, uint8_t* out, unsigned cap)
{
	cimbar::zstd_decompressor<std::stringstream> d;
	if (!d.write((const char*)in, n)) return -1;
	const std::string s = d.str();
	if (s.size() > cap) return -2;
	std::memcpy(out, s.data(), s.size());
	return (int)s.size();
}

// the fountain chunk stream itself (what the frames above carry), 625 bytes per chunk
int ref_fountain_chunks(const uint8_t* data, unsigned size, int encode_id, unsigned nchunks, uint8_t* out)
{
	std::stringstream ss(std::string((const char*)data, size));
	Encoder enc;
	enc.set_encode_id((uint8_t)encode_id);
	fountain_encoder_stream::ptr fes = enc.create_fountain_encoder(ss, "", 0);
	if (!fes) return -1;
	unsigned cs = cimbar::Config::fountain_chunk_size();
	for (unsigned c = 0; c < nchunks; ++c)
	{
		if (fes->readsome((char*)out + (size_t)c * cs, cs) != (std::streamsize)cs) return (int)c;
	}
	return (int)nchunks;
}

void ref_reset_ccm() { reset_thread_ccm(); }

int ref_get_ccm(float* out9)
{
	TestableCimbDecoder d(cimbar::Config::symbol_bits(), cimbar::Config::color_bits(), cimbar::Config::dark(), 0xFF);
	const color_correction& cc = d.get_ccm();
	cv::Matx<float, 3, 3> m = cc.mat();
	for (int i = 0; i < 9; ++i) out9[i] = m.val[i];
	return cc.active() ? 1 : 0;
}

// Decoder::decode_fountain on one RGB8 frame. chunks: 12*625 bytes, slot j = fountain chunk j of the frame (zeros if dropped).
// Returns the reference's return value (good bytes). reset_ccm!=0 clears the thread_local CCM first.
int ref_decode_fountain(const uint8_t* rgb, unsigned w, unsigned h, int preprocess, int color_correction, int reset_ccm,
                        uint8_t* chunks, uint32_t* good_mask)
{
	if (reset_ccm) reset_thread_ccm();
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	unsigned cs = cimbar::Config::fountain_chunk_size();
	unsigned per_frame = cimbar::Config::fountain_chunks_per_frame(cimbar::Config::bits_per_cell());

	RefDecoder dec;
	chunk_collector col(cs);
	std::vector<unsigned> slots;
	unsigned res = dec.decode_fountain_masked(img, col, slots, preprocess != 0, color_correction);

	std::memset(chunks, 0, (size_t)per_frame * cs);
	uint32_t mask = 0;
	for (size_t k = 0; k < slots.size(); ++k)
	{
		if (slots[k] >= per_frame) continue;
		std::memcpy(chunks + (size_t)slots[k] * cs, col.bytes.data() + k * cs, cs);
		mask |= 1u << slots[k];
	}
	if (good_mask) *good_mask = mask;
	return (int)res;
}

// Decoder::decode (Decoder.h:163-169) on one RGB8 frame into a std::stringstream, the way `./cimbar --no-fountain` writes its
// output file (cimbar.cpp:270-272): 60 RS outputs back to back, a failed block as 125 zero bytes. Returns the reference's return
// value (the stream's tellp after the frame); bytes must hold 7500.
int ref_decode_plain(const uint8_t* rgb, unsigned w, unsigned h, int preprocess, int color_correction, int reset_ccm, uint8_t* bytes)
{
	if (reset_ccm) reset_thread_ccm();
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	Decoder dec;
	std::stringstream ss;
	unsigned res = dec.decode(img, ss, preprocess != 0, color_correction);
	const std::string out = ss.str();
	std::memset(bytes, 0, 7500);
	std::memcpy(bytes, out.data(), std::min<size_t>(out.size(), 7500));
	return (int)res;
}

// The literal public entry point, writing into an escrow_buffer_writer exactly like cimbard_scan_extract_decode
// (cimbar_recv_js.cpp:160-188). Returns buffers_in_use()*625; chunks packed contiguously.
int ref_decode_fountain_escrow(const uint8_t* rgb, unsigned w, unsigned h, int preprocess, int color_correction, int reset_ccm,
                               uint8_t* bufspace)
{
	if (reset_ccm) reset_thread_ccm();
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	unsigned cs = cimbar::Config::fountain_chunk_size();
	unsigned per_frame = cimbar::Config::fountain_chunks_per_frame(cimbar::Config::bits_per_cell());
	escrow_buffer_writer ebw(bufspace, per_frame, cs);
	Decoder dec;
	dec.decode_fountain(img, ebw, preprocess != 0, color_correction);
	return (int)(ebw.buffers_in_use() * cs);
}

// CimbReader internals for stage-level pinning: the packed bitplane (CimbReader.cpp:30-46), then the flood-ordered
// symbol pass (CimbReader.cpp:139-162): per visit k -> cell index, drifted x, y, symbol bits.
int ref_symbol_pass(const uint8_t* rgb, unsigned w, unsigned h, int preprocess, uint8_t* bitplane /* w*h/8 or NULL */,
                    int32_t* visit /* 4 ints per cell or NULL */)
{
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	TestableCimbDecoder d(cimbar::Config::symbol_bits(), cimbar::Config::color_bits(), cimbar::Config::dark(), 0xFF);
	ExposedReader reader(img, d, cimbar::Config::color_mode(), preprocess != 0, 0);
	if (bitplane) std::memcpy(bitplane, reader.bitplane().data(), std::min<size_t>(reader.bitplane().size(), (size_t)w * h / 8));
	int k = 0;
	while (!reader.done())
	{
		PositionData pos;
		unsigned bits = reader.read(pos);
		if (visit) { visit[4*k] = (int)pos.i; visit[4*k+1] = pos.x; visit[4*k+2] = pos.y; visit[4*k+3] = (int)bits; }
		++k;
	}
	return k;
}

// colour classifier on explicit inputs (CimbDecoder.cpp:168-200), using the thread's current CCM
int ref_best_color(float r, float g, float b)
{
	TestableCimbDecoder d(cimbar::Config::symbol_bits(), cimbar::Config::color_bits(), cimbar::Config::dark(), 0xFF);
	return (int)d.get_best_color(r, g, b, cimbar::Config::color_mode());
}

// libcorrect through the reference's wrapper (ReedSolomon.h:23-47)
int ref_rs_encode(const uint8_t* msg, unsigned msg_len, unsigned parity, uint8_t* out)
{
	ReedSolomon rs(parity);
	return (int)rs.encode((const char*)msg, msg_len, (char*)out);
}
int ref_rs_decode(const uint8_t* enc, unsigned enc_len, unsigned parity, uint8_t* out)
{
	static thread_local std::unique_ptr<ReedSolomon> rs;
	static thread_local unsigned rs_parity = 0;
	if (!rs || rs_parity != parity) { rs.reset(new ReedSolomon(parity)); rs_parity = parity; }
	return (int)rs->decode((const char*)enc, enc_len, (char*)out);
}

int ref_interleave_reverse(unsigned size, unsigned blocks, unsigned partitions, uint32_t* out)
{
	std::vector<unsigned> v = Interleave::interleave_reverse(size, blocks, partitions);
	for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
	return (int)v.size();
}

// fountain_decoder_sink (fountain_decoder_sink.h:53-214) -- the rank-0 wirehair sink of BASELINE config 4
int ref_sink_reset(unsigned chunk_size) { g_sink = std::make_shared<fountain_decoder_sink>(chunk_size); return 0; }
int64_t ref_sink_decode_frame(const uint8_t* buf, unsigned size) { return g_sink ? g_sink->decode_frame((const char*)buf, size) : -100; }
// n frames' worth of chunk slots fed in frame order then chunk order (what multigpu.feed_sink does call by call); a finished stream is
// recovered into `out` at once (which also marks it done, so later chunks of it are ignored: fountain_decoder_sink.h:77-96,146-148).
// Returns the number of chunks handed to the sink; *completed_id = the file id once complete (else 0).
int64_t ref_sink_feed_batch(const uint8_t* chunks, const uint32_t* masks, unsigned n, unsigned chunks_per_frame, unsigned chunk_size, uint8_t* out,
                            unsigned out_size, uint32_t* completed_id)
{
	if (!g_sink) return -100;
	int64_t fed = 0;
	for (unsigned f = 0; f < n; ++f)
		for (unsigned j = 0; j < chunks_per_frame; ++j)
		{
			if (!(masks[f] & (1u << j))) continue;
			++fed;
			int64_t r = g_sink->decode_frame((const char*)chunks + ((size_t)f * chunks_per_frame + j) * chunk_size, chunk_size);
			if (r > 0)
			{
				if (out && g_sink->recover((uint32_t)r, out, out_size) && completed_id) *completed_id = (uint32_t)r;
			}
		}
	return fed;
}
// The same, stopping at the chunk that completes the file (the sink ignores everything after it anyway: is_done) and saying where the time
// went: wirehair takes blocks one by one and, on the block that makes the system solvable, runs its solve inside that call -- `solve_s` is the
// duration of that one call (+ recover), `feed_s` the sum of all the others. bench --config 4 reports them separately.
int64_t ref_sink_feed_batch_timed(const uint8_t* chunks, const uint32_t* masks, unsigned n, unsigned chunks_per_frame, unsigned chunk_size, uint8_t* out,
                                  unsigned out_size, uint32_t* completed_id, double* feed_s, double* solve_s)
{
	if (!g_sink) return -100;
	int64_t fed = 0;
	double feed = 0, solve = 0;
	for (unsigned f = 0; f < n; ++f)
		for (unsigned j = 0; j < chunks_per_frame; ++j)
		{
			if (!(masks[f] & (1u << j))) continue;
			++fed;
			const auto t0 = std::chrono::steady_clock::now();
			int64_t r = g_sink->decode_frame((const char*)chunks + ((size_t)f * chunks_per_frame + j) * chunk_size, chunk_size);
			bool done = false;
			if (r > 0 && out && g_sink->recover((uint32_t)r, out, out_size)) { done = true; if (completed_id) *completed_id = (uint32_t)r; }
			const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (r > 0) solve += dt; else feed += dt;
			if (done) { if (feed_s) *feed_s += feed; if (solve_s) *solve_s += solve; return fed; }
		}
	if (feed_s) *feed_s += feed;
	if (solve_s) *solve_s += solve;
	return fed;
}
int ref_sink_is_done(uint32_t id) { return g_sink && g_sink->is_done(id) ? 1 : 0; }
int ref_sink_recover(uint32_t id, uint8_t* out, unsigned size) { return g_sink && g_sink->recover(id, out, size) ? 1 : 0; }


// ---- the stage in front of the decoder: Scanner / Deskewer / Extractor on one RGB8 camera frame
// Scanner::preprocess_image(img, fast=true) (Scanner.h:148-165): gray -> GaussianBlur -> Otsu threshold. out: w*h bytes (0 / 255)
int ref_scan_preprocess(const uint8_t* rgb, unsigned w, unsigned h, uint8_t* out)
{
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	cv::Mat bin = Scanner::preprocess_image(img, true);
	for (int y = 0; y < bin.rows; ++y) std::memcpy(out + (size_t)y * w, bin.ptr<uchar>(y), w);
	return 0;
}

// Scanner(img).scan() + Corners (Extractor.h:33-39): corners[8] = top_left, top_right, bottom_left, bottom_right (x, y). Returns the
// number of anchors found (4 = success).
int ref_scan_corners(const uint8_t* rgb, unsigned w, unsigned h, float* corners8)
{
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	Scanner scanner(img);
	std::vector<Anchor> points = scanner.scan();
	if (points.size() < 4) return (int)points.size();
	Corners corners(points);
	std::vector<cv::Point2f> all = corners.all();
	for (int i = 0; i < 4; ++i) { corners8[2 * i] = all[i].x; corners8[2 * i + 1] = all[i].y; }
	return 4;
}

// Scanner(img).scan() raw: up to 4 anchors as {x, xmax, y, ymax}; returns how many scan() produced
int ref_scan_anchors(const uint8_t* rgb, unsigned w, unsigned h, int32_t* out16)
{
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	Scanner scanner(img);
	std::vector<Anchor> points = scanner.scan();
	for (size_t i = 0; i < points.size() && i < 4; ++i)
	{
		out16[4 * i] = points[i].x(); out16[4 * i + 1] = points[i].xmax(); out16[4 * i + 2] = points[i].y(); out16[4 * i + 3] = points[i].ymax();
	}
	return (int)points.size();
}

// Deskewer(0, image_size, anchor_size).deskew(img, corners) (Deskewer.h:26-40) for explicit corners: out = 1024*1024*3
int ref_deskew(const uint8_t* rgb, unsigned w, unsigned h, const float* corners8, uint8_t* out)
{
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	Corners corners(point<int>((int)corners8[0], (int)corners8[1]), point<int>((int)corners8[2], (int)corners8[3]),
	                point<int>((int)corners8[4], (int)corners8[5]), point<int>((int)corners8[6], (int)corners8[7]));
	Deskewer de;
	cv::Mat res = de.deskew(img, corners);
	for (int y = 0; y < res.rows; ++y) std::memcpy(out + (size_t)y * res.cols * 3, res.ptr<uchar>(y), (size_t)res.cols * 3);
	return res.rows;
}

// Extractor::extract (Extractor.h:29-45): 0 failure, 1 success, 2 needs sharpen; out = the deskewed 1024x1024 RGB8 frame
int ref_extract(const uint8_t* rgb, unsigned w, unsigned h, uint8_t* out)
{
	cv::Mat img((int)h, (int)w, CV_8UC3, (void*)rgb);
	cv::Mat res;
	Extractor ext;
	int rc = ext.extract(img, res);
	if (rc != Extractor::FAILURE)
		for (int y = 0; y < res.rows; ++y) std::memcpy(out + (size_t)y * res.cols * 3, res.ptr<uchar>(y), (size_t)res.cols * 3);
	return rc;
}


// The body of CimbDecoderTest/testPrethresholdDecode (cimb_translator/test/CimbDecoderTest.cpp:49-75), run against the cv-shim: every tile,
// embedded in a 10x10 window, goes through cvtColor(RGB2GRAY) + adaptiveThreshold(MEAN_C, blockSize 9, C 0) + mat_to_bitbuffer and must
// decode to itself at the centre with distance 0 -- which the reference's CI establishes for a real OpenCV. Returns the number of tiles
// (0..16) for which the shim reproduces that; out3[3*i..] = symbol, drift_offset, distance.
int ref_prethreshold_decode_test(unsigned* out3)
{
	CimbDecoder cd(4, 0, true, 0xFF);
	int good = 0;
	for (unsigned i = 0; i < 16; ++i)
	{
		cv::Mat tile = cimbar::getTile(4, i, true);
		cv::Mat tenxten(10, 10, tile.type(), cv::Scalar(0, 0, 0));
		tile.copyTo(tenxten(cv::Rect(1, 1, tile.cols, tile.rows)));
		cv::cvtColor(tenxten, tenxten, cv::COLOR_RGB2GRAY);
		cv::adaptiveThreshold(tenxten, tenxten, 255, cv::ADAPTIVE_THRESH_MEAN_C, cv::THRESH_BINARY, 9, 0);
		bitbuffer bb((100 / 8) + 1);
		bitmatrix::mat_to_bitbuffer(tenxten, bb.get_writer());
		bitmatrix bm(bb, 10, 10);
		unsigned drift_offset = 99, distance = 99;
		unsigned res = cd.decode_symbol(bm, drift_offset, distance);
		if (out3) { out3[3 * i] = res; out3[3 * i + 1] = drift_offset; out3[3 * i + 2] = distance; }
		good += (res == i && drift_offset == 4 && distance == 0) ? 1 : 0;
	}
	return good;
}

// cv::cvtColor of the cv-shim on a caller-described matrix (rows x cols x channels, 8-bit) -- the conversions get_rgb makes
// (cimbar_js/cimbar_recv_js.cpp:94-120: COLOR_YUV2RGB_NV12 = 90, COLOR_YUV420p2RGB = 98, COLOR_RGBA2RGB = 1); out = rows_out x cols x 3.
// The whole of get_rgb -> Extractor -> Decoder is reachable as the reference's own cimbard_scan_extract_decode, linked into this library.
int ref_cvtcolor(const uint8_t* src, int rows, int cols, int channels, int code, uint8_t* out)
{
	cv::Mat m(rows, cols, CV_MAKETYPE(CV_8U, channels), (void*)src);
	cv::Mat dst;
	cv::cvtColor(m, dst, code);
	for (int y = 0; y < dst.rows; ++y) std::memcpy(out + (size_t)y * dst.cols * dst.channels(), dst.ptr<uchar>(y), (size_t)dst.cols * dst.channels());
	return dst.rows;
}

// cv::cvtColor(RGB2GRAY) + cv::GaussianBlur(unit x unit, 0) of the cv-shim, the first two calls of Scanner::preprocess_image (Scanner.h:151-160), with the
// kernel size given by the caller: the blur by itself, for the OpenCV pin vectors (tests/test_opencv_pin_vectors.py)
int ref_gray_blur(const uint8_t* rgb, int w, int h, int unit, uint8_t* out)
{
	cv::Mat img(h, w, CV_8UC3, (void*)rgb), gray;
	cv::cvtColor(img, gray, cv::COLOR_RGB2GRAY);
	cv::GaussianBlur(gray, gray, cv::Size(unit, unit), 0);
	for (int y = 0; y < h; ++y) std::memcpy(out + (size_t)y * w, gray.ptr<uchar>(y), (size_t)w);
	return unit;
}

// cv::threshold(BINARY | OTSU) of the cv-shim on a gray image (Scanner::threshold_fast, Scanner.h:126-130, drops the value it returns): out = 0 / 255,
// returns the threshold
int ref_otsu_threshold(const uint8_t* gray, int w, int h, uint8_t* out)
{
	cv::Mat g(h, w, CV_8UC1, (void*)gray), bin;
	const double t = cv::threshold(g, bin, 0, 255, cv::THRESH_BINARY | cv::THRESH_OTSU);
	for (int y = 0; y < h; ++y) std::memcpy(out + (size_t)y * w, bin.ptr<uchar>(y), (size_t)w);
	return (int)t;
}

// color_correction::get_moore_penrose_lsm (chromatic_adaptation/color_correction.h:26-39: transpose, invert(DECOMP_SVD), product) on caller-described
// rows x 3 float matrices, built the way CimbReader::init_ccm builds them (CimbReader.cpp:229-263: push_back of 1x3 rows): the one cv::invert the path makes
int ref_moore_penrose_lsm(const float* actual, const float* desired, int rows, float* out9)
{
	cv::Mat a = cv::Mat::ones(0, 3, CV_32F), d = cv::Mat::ones(0, 3, CV_32F);
	for (int r = 0; r < rows; ++r) {
		cv::Mat arow = (cv::Mat_<float>(1, 3) << actual[3 * r], actual[3 * r + 1], actual[3 * r + 2]);
		cv::Mat drow = (cv::Mat_<float>(1, 3) << desired[3 * r], desired[3 * r + 1], desired[3 * r + 2]);
		a.push_back(arow);
		d.push_back(drow);
	}
	const cv::Matx<float, 3, 3> m = color_correction::get_moore_penrose_lsm(a, d);
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out9[3 * i + j] = m(i, j);
	return rows;
}

}  // extern "C"
