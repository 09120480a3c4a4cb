// cimbar_amd::Decoder -- C++ host adapter over the C ABI (include/cimbar_hip.h), mirroring the reference's frame-codec
// surface for the decode_fountain path so that it drops in behind ./cimbar:
//
//     reference                                                  here
//     ---------------------------------------------------------  ---------------------------------------------------
//     class Decoder            src/lib/encoder/Decoder.h:16-38   cimbar_amd::Decoder           (same ctor / decode_fountain signature)
//     STREAM concept           Decoder.h:171-189,                any type with chunk_size() and write(const char*, unsigned):
//                              reed_solomon_stream.h:91-114      fountain_decoder_sink, concurrent_fountain_decoder_sink,
//                                                                escrow_buffer_writer, null_stream -- used UNCHANGED
//     class CimbReader         src/lib/cimb_translator/          cimbar_amd::CimbReader        (read / read_color / done / num_reads over the
//                              CimbReader.h:13-41                                               GPU's per-cell results, linear cell order)
//     class CimbDecoder        cimb_translator/CimbDecoder.h     cimbar_amd::CimbDecoder       (what CimbReader's constructor takes; owns nothing, names the context)
//     class Deskewer           src/lib/extractor/Deskewer.h:12-40 cimbar_amd::Deskewer          (deskew(img, corners) -> 1024x1024 frame; plus
//     Scanner::preprocess_image src/lib/extractor/Scanner.h:148-165                              scan_preprocess(img) -> the binary image Scanner scans)
//     class Extractor          src/lib/extractor/Extractor.h:11-45 cimbar_amd::Extractor        (extract(img, out) -> FAILURE / SUCCESS / NEEDS_SHARPEN, the
//                                                                                               anchor search included, on the device)
//
//     class DecoderPlus        src/lib/encoder/DecoderPlus.h:11-58 cimbar_amd::Decoder::load_ccm / save_ccm (`--color-correction-file`, cimbar.cpp:265-266,297-298)
//
// Header-only; link against libcimbar_hip.so. No exceptions, no OpenCV requirement: MAT is anything shaped like cv::Mat
// (`data`, `cols`, `rows`, `step`), e.g. cv::Mat or cimbar_amd::image_view below -- or anything with a `getMat(access)` member that
// returns such a thing: cv::UMat as it is, which is what ./cimbar hands to decode_fountain and Extractor::extract (cimbar.cpp:132,146,167-171;
// CimbReader.h:16-17 has the same pair of constructors).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/cimbar_hip.h"

namespace cimbar_amd {

// a borrowed RGB8 image, for callers without OpenCV
struct image_view
{
	const unsigned char* data = nullptr;
	int cols = 0;
	int rows = 0;
	size_t step = 0;   // bytes per row; 0 means cols*3
};

struct PositionData   // src/lib/cimb_translator/PositionData.h
{
	unsigned i = 0;
	int x = 0;
	int y = 0;
};

// cv::UMat and friends: `img.getMat(cv::ACCESS_READ)` without naming OpenCV. The access argument's type is taken from the member's own
// signature (an enum in OpenCV 4, an int in OpenCV 3) and its value is ACCESS_READ = 1 << 24 in both (core/mat.hpp, enum AccessFlag).
namespace detail {
template <typename T> struct getmat_arg;
template <typename C, typename R, typename A> struct getmat_arg<R (C::*)(A) const> { typedef A type; };
template <typename T, typename = void> struct has_getmat : std::false_type {};
template <typename T> struct has_getmat<T, decltype(void(&T::getMat))> : std::true_type {};

// `use(m)` is called with something that has data / cols / rows / step: `img` itself, or the cv::Mat header img.getMat(ACCESS_READ) returns
// (alive for the duration of the call, like the temporary in CimbReader's UMat constructor, CimbReader.cpp:128-131). Tag dispatch, so that
// `use` is only ever instantiated with the type it will really see.
template <typename MAT, typename F>
auto with_mat_impl(const MAT& img, F&& use, std::false_type) { return use(img); }
template <typename MAT, typename F>
auto with_mat_impl(const MAT& img, F&& use, std::true_type)
{
	typedef typename getmat_arg<decltype(&MAT::getMat)>::type access_t;
	const auto m = img.getMat(static_cast<access_t>(1 << 24));
	return use(m);
}
template <typename MAT, typename F>
auto with_mat(const MAT& img, F&& use) { return with_mat_impl(img, std::forward<F>(use), has_getmat<MAT>()); }
// the same for an image that is written: getMat(ACCESS_RW = 3 << 24) maps the UMat's buffer, the header's destructor unmaps it
template <typename MAT, typename F>
void with_mat_rw_impl(MAT& img, F&& use, std::false_type) { use(img); }
template <typename MAT, typename F>
void with_mat_rw_impl(MAT& img, F&& use, std::true_type)
{
	typedef typename getmat_arg<decltype(&MAT::getMat)>::type access_t;
	auto m = img.getMat(static_cast<access_t>(3 << 24));
	use(m);
}
template <typename MAT, typename F>
void with_mat_rw(MAT& img, F&& use) { with_mat_rw_impl(img, std::forward<F>(use), has_getmat<MAT>()); }
}  // namespace detail

class Decoder
{
public:
	// Decoder(use_ecc, interleave) as in Decoder.h:40-45. Only the reference's defaults (ECC on, interleave on) exist on the GPU path, in every
	// mode Config::temp_conf lists (68 "B", 67 "Bm", 66 "Bu", the legacy 4 and 8; any other value is mode B there and here, Config.h:41-43) --
	// what cimbard_configure_decode / Config::update(mode_val) selects in the reference. ECC or interleave off leaves the object !good() and
	// every decode returns 0, like a reference decode that found nothing.
	explicit Decoder(bool use_ecc = true, bool interleave = true, int device = 0, int mode_val = 68)
	{
		if (use_ecc && interleave) _rc = cimbar_hip_create(device, mode_val, &_ctx);
		else _rc = CIMBAR_HIP_EINVAL;
		if (_ctx && cimbar_hip_geometry(_ctx, _geo) != CIMBAR_HIP_GEOMETRY_WORDS) { cimbar_hip_destroy(_ctx); _ctx = nullptr; _rc = CIMBAR_HIP_EHIP; }
		if (_ctx) _frame.resize((size_t)cimbar_hip_ctx_bufsize(_ctx));   // one frame's chunk space, by the CONTEXT's mode (8750 bytes in mode 8)
	}
	~Decoder() { if (_ctx) cimbar_hip_destroy(_ctx); }
	Decoder(const Decoder&) = delete;
	Decoder& operator=(const Decoder&) = delete;

	bool good() const { return _ctx != nullptr; }
	int error_code() const { return _rc; }
	const char* last_error() const { return cimbar_hip_last_error(_ctx); }
	cimbar_hip_ctx* context() { return _ctx; }

	// the Config:: getters of the context's mode (Config.h:52-165). The CIMBAR_HIP_CHUNK_SIZE / _FRAME_BYTES macros of cimbar_hip.h are mode
	// B's values and NOT the largest (mode 8: 10 x 875 = 8750 bytes): size by frame_bytes() / cimbar_hip_ctx_bufsize, or CIMBAR_HIP_MAX_FRAME_BYTES.
	int mode() const { return (int)_geo[0]; }
	unsigned image_size_x() const { return (unsigned)_geo[1]; }
	unsigned image_size_y() const { return (unsigned)_geo[2]; }
	unsigned total_cells() const { return (unsigned)_geo[3]; }
	unsigned fountain_chunks_per_frame() const { return (unsigned)_geo[4]; }
	unsigned fountain_chunk_size() const { return (unsigned)_geo[5]; }
	unsigned frame_bytes() const { return (unsigned)(_geo[4] * _geo[5]); }
	unsigned cells_per_col_x() const { return (unsigned)_geo[9]; }
	unsigned cells_per_col_y() const { return (unsigned)_geo[10]; }
	unsigned cell_offset() const { return (unsigned)_geo[11]; }
	unsigned symbol_bits() const { return 4; }                                       // every 8x8 configuration (GridConf.h:126,149,173)
	unsigned color_bits() const { return mode() == 8 ? 3u : 2u; }                    // Config.h:24-35
	unsigned color_mode() const { return (mode() == 4 || mode() == 8) ? 0u : 1u; }   // Config::color_mode(): legacy_mode ? 0 : 1 (Config.h:61-64)

	// DecoderPlus::load_ccm / save_ccm (DecoderPlus.h:32-58): a file of nine floats, row-major 3x3 -- `./cimbar --color-correction-file`
	// loads it before a --no-fountain decode and saves the matrix in force after a fountain decode (cimbar.cpp:265-266,297-298).
	bool update_color_correction(const float m9[9]) { return _ctx && (_rc = cimbar_hip_set_ccm(_ctx, m9)) == 0; }   // CimbDecoder::update_color_correction
	bool load_ccm(const std::string& filename)
	{
		float m[9];
		FILE* f = std::fopen(filename.c_str(), "rb");
		if (!f) return false;
		const size_t got = std::fread(m, 1, sizeof m, f);
		std::fclose(f);
		if (got < sizeof m) return false;            // `data.size() < 3*3*4` (DecoderPlus.h:36-37)
		return update_color_correction(m);
	}
	bool save_ccm(const std::string& filename)
	{
		float m[9];
		if (!_ctx || cimbar_hip_get_ccm(_ctx, m) != 1) return false;   // `not get_ccm().active()` (DecoderPlus.h:49-50)
		FILE* f = std::fopen(filename.c_str(), "wb");
		if (!f) return false;
		const size_t put = std::fwrite(m, 1, sizeof m, f);
		std::fclose(f);
		return put == sizeof m;
	}

	// Decoder::decode_fountain (Decoder.h:171-189): good chunks reach ostream.write(buf, 625) in chunk order, exactly what
	// aligned_stream would have delivered; returns the cumulative good bytes. A sink whose chunk_size() is not 625 receives
	// nothing but the byte count is still returned (Decoder.h:180-185).
	template <typename MAT, typename FOUNTAINSTREAM>
	unsigned decode_fountain(const MAT& img, FOUNTAINSTREAM& ostream, bool should_preprocess = false, int color_correction = 2)
	{
		return detail::with_mat(img, [&](const auto& m) { return decode_fountain_mat(m, ostream, should_preprocess, color_correction); });
	}

	// The same with ONE frame in flight, for cimbar.cpp:124-171's loop (the next image is read / extracted while this one decodes): the call starts
	// this frame on the device (cimbar_hip_decode_frame_async) and delivers -- to the SAME sink, in frame order -- the chunks of the frame started by
	// the call before it, returning THAT frame's byte count (0 for the very first call). `flush(sink)` after the loop delivers the last frame.
	// The image may be reused or freed as soon as the call returns unless it lies in page-locked memory, which must stay until the next call / flush.
	// Chunk order at the sink, the colour-correction carry-over and the sum of the return values are those of decode_fountain.
	template <typename MAT, typename FOUNTAINSTREAM>
	unsigned decode_fountain_overlapped(const MAT& img, FOUNTAINSTREAM& ostream, bool should_preprocess = false, int color_correction = 2)
	{
		return detail::with_mat(img, [&](const auto& m) { return decode_fountain_overlapped_mat(m, ostream, should_preprocess, color_correction); });
	}
	template <typename FOUNTAINSTREAM>
	unsigned flush(FOUNTAINSTREAM& ostream)
	{
		if (!_ctx || _inflight < 0) return 0;
		const long long t = _inflight;
		_inflight = -1;
		const int res = cimbar_hip_decode_frame_wait(_ctx, t);      // writes the slot's chunks and mask
		return deliver(res, _ov[_inflight_slot].chunks.data(), _ov[_inflight_slot].mask, ostream);
	}

	// Decoder::decode (Decoder.h:163-169), the `--no-fountain` path (cimbar.cpp:270-272): the frame's 60 Reed-Solomon outputs go to
	// ostream.write back to back, a block that could not be decoded as 125 zero bytes (what reed_solomon_stream does for an
	// ofstream / stringstream, reed_solomon_stream.h:96-107). Returns ostream.tellp() like the reference (Decoder.h:116-117).
	template <typename MAT, typename STREAM>
	unsigned decode(const MAT& img, STREAM& ostream, bool should_preprocess = false, int color_correction = 2)
	{
		return detail::with_mat(img, [&](const auto& m) { return decode_mat(m, ostream, should_preprocess, color_correction); });
	}

	// n densely packed image_size_x x image_size_y RGB8 frames in host memory, decoded on the GPU in one batch; chunks go to the sink in frame
	// order then chunk order (what a single-threaded reference loop over the frames would have produced). Returns total good bytes.
	template <typename FOUNTAINSTREAM>
	unsigned long long decode_fountain_batch(const unsigned char* frames, int n, FOUNTAINSTREAM& ostream, bool should_preprocess = false,
	                                         int color_correction = 2)
	{
		if (!_ctx || n <= 0) return 0;
		const unsigned cs = fountain_chunk_size(), per = fountain_chunks_per_frame();
		_chunks.resize((size_t)n * cs * per);
		_masks.resize((size_t)n);
		int64_t res = cimbar_hip_decode_batch(_ctx, frames, n, CIMBAR_HIP_MEM_HOST, should_preprocess ? 1 : 0, color_correction,
		                                      _chunks.data(), _masks.data(), CIMBAR_HIP_MEM_HOST, nullptr);
		_rc = res < 0 ? (int)res : 0;
		if (res <= 0) return 0;
		if (ostream.chunk_size() == cs)
			for (int f = 0; f < n; ++f)
				for (unsigned j = 0; j < per; ++j)
					if (_masks[f] & (1u << j)) ostream.write(reinterpret_cast<const char*>(_chunks.data()) + ((size_t)f * per + j) * cs, cs);
		return (unsigned long long)res;
	}

	const std::vector<uint32_t>& last_masks() const { return _masks; }

protected:
	// what decode_fountain does with one frame's result: good chunks to the sink in chunk order, the byte count back
	template <typename FOUNTAINSTREAM>
	unsigned deliver(int res, const unsigned char* chunks, uint32_t mask, FOUNTAINSTREAM& ostream)
	{
		_rc = res < 0 ? res : 0;
		if (res <= 0) return 0;   // CimbReader::_good == false / nothing decoded
		const unsigned cs = fountain_chunk_size();
		if (ostream.chunk_size() == cs)
			for (unsigned j = 0; j < fountain_chunks_per_frame(); ++j)
				if (mask & (1u << j)) ostream.write(reinterpret_cast<const char*>(chunks) + (size_t)j * cs, cs);
		return (unsigned)res;
	}

	template <typename MAT, typename FOUNTAINSTREAM>
	unsigned decode_fountain_mat(const MAT& img, FOUNTAINSTREAM& ostream, bool should_preprocess, int color_correction)
	{
		if (!_ctx) return 0;
		// (a frame left in flight by decode_fountain_overlapped goes first: frame order at the sink. Its byte count is part of what this call returns --
		// the sum of the return values stays the plain loop's -- and an error it reported is not forgotten over this frame's success)
		const unsigned flushed = flush(ostream);
		const int flushed_rc = _rc;
		unsigned char* chunks = _frame.data();   // frame_bytes() of the context's mode
		uint32_t mask = 0;
		const size_t step = image_step(img);
		int res = cimbar_hip_decode_frame(_ctx, reinterpret_cast<const uint8_t*>(img.data), (unsigned)img.cols, (unsigned)img.rows, step,
		                                  should_preprocess ? 1 : 0, color_correction, chunks, &mask);
		const unsigned own = deliver(res, chunks, mask, ostream);
		if (flushed_rc < 0 && _rc == 0) _rc = flushed_rc;
		return flushed + own;
	}

	template <typename MAT, typename FOUNTAINSTREAM>
	unsigned decode_fountain_overlapped_mat(const MAT& img, FOUNTAINSTREAM& ostream, bool should_preprocess, int color_correction)
	{
		if (!_ctx) return 0;
		// this frame is started first -- its copy and kernels run while the previous frame's chunks go through the sink below -- into the slot
		// the previous frame does not use (the library writes through the pointers it is given here when the frame is waited for)
		const int slot = _inflight >= 0 ? 1 - _inflight_slot : 0;
		_ov[slot].chunks.resize(_frame.size());
		const long long t = cimbar_hip_decode_frame_async(_ctx, reinterpret_cast<const uint8_t*>(img.data), (unsigned)img.cols, (unsigned)img.rows, image_step(img),
		                                                  should_preprocess ? 1 : 0, color_correction, _ov[slot].chunks.data(), &_ov[slot].mask);
		const unsigned delivered = flush(ostream);
		if (t < 0) { _rc = (int)t; return delivered; }
		_inflight = t;
		_inflight_slot = slot;
		return delivered;
	}

	template <typename MAT, typename STREAM>
	unsigned decode_mat(const MAT& img, STREAM& ostream, bool should_preprocess, int color_correction)
	{
		if (!_ctx) return 0;
		if ((unsigned)img.cols < image_size_x() || (unsigned)img.rows < image_size_y()) {
			// CimbReader::_good == false (CimbReader.cpp:119): the reference writes its all-zero Reed-Solomon outputs
			const std::vector<char> z(frame_bytes(), 0);
			ostream.write(z.data(), frame_bytes());
			return (unsigned)ostream.tellp();
		}
		if ((unsigned)img.cols != image_size_x() || (unsigned)img.rows != image_size_y()) { _rc = CIMBAR_HIP_EDIM; return 0; }   // (padded images: decode_fountain only)
		std::vector<unsigned char> packed;
		const unsigned char* src = reinterpret_cast<const unsigned char*>(img.data);
		const size_t step = image_step(img), dense = (size_t)image_size_x() * 3;
		if (step != dense) {
			packed.resize(dense * image_size_y());
			for (int y = 0; y < (int)image_size_y(); ++y)
				for (size_t k = 0; k < dense; ++k) packed[(size_t)y * dense + k] = src[(size_t)y * step + k];
			src = packed.data();
		}
		unsigned char* bytes = _frame.data();   // frame_bytes() of the context's mode
		int64_t res = cimbar_hip_decode_plain_batch(_ctx, src, 1, CIMBAR_HIP_MEM_HOST, should_preprocess ? 1 : 0, color_correction, bytes, nullptr,
		                                            CIMBAR_HIP_MEM_HOST, nullptr);
		_rc = res < 0 ? (int)res : 0;
		if (res <= 0) return 0;
		ostream.write(reinterpret_cast<const char*>(bytes), frame_bytes());
		return (unsigned)ostream.tellp();
	}

	template <typename MAT>
	static size_t image_step(const MAT& img)
	{
		size_t s = (size_t)img.step;
		return s ? s : (size_t)img.cols * 3;
	}

	cimbar_hip_ctx* _ctx = nullptr;
	int _rc = 0;
	int32_t _geo[CIMBAR_HIP_GEOMETRY_WORDS] = {};
	std::vector<unsigned char> _frame;    // one frame's chunk space (cimbar_hip_ctx_bufsize)
	struct overlap_slot { std::vector<unsigned char> chunks; uint32_t mask = 0; } _ov[2];   // decode_fountain_overlapped: the frame in flight and the one being started
	long long _inflight = -1;
	int _inflight_slot = 0;
	std::vector<unsigned char> _chunks;
	std::vector<uint32_t> _masks;
};

// CimbReader's cell-level surface (CimbReader.h:13-41) over the GPU results of ONE frame. The frame is decoded at
// construction; read() then walks the cells in linear index order (the reference walks them in flood order, but its caller
// Decoder::do_decode places every result by pos.i, Decoder.h:84-97, so the order is not observable there) and returns the
// 4 symbol bits plus the drifted position the colour pass used; read_color() returns the 2 colour bits of that cell.
// CimbDecoder as CimbReader's constructor wants it (cimb_translator/CimbDecoder.h: CimbDecoder(symbol_bits, color_bits, dark, ahashThreshold)):
// the tile hashes and the colour-correction state live in the device context, so this only names one; the bit counts must be the context's
// mode's -- what the reference's callers pass anyway: CimbDecoder(Config::symbol_bits(), Config::color_bits()) (Decoder.h:40-45): 4 + 2, 4 + 3 in mode 8.
class CimbDecoder
{
public:
	explicit CimbDecoder(Decoder& decoder, unsigned symbol_bits = 4, unsigned color_bits = 2, bool dark = true, unsigned char ahash_threshold = 0xFF)
	    : _dec(decoder), _ok(symbol_bits == decoder.symbol_bits() && color_bits == decoder.color_bits() && dark)
	{
		(void)ahash_threshold;
	}
	bool good() const { return _ok && _dec.good(); }
	Decoder& decoder() { return _dec; }

protected:
	Decoder& _dec;
	bool _ok;
};

class CimbReader
{
public:
	// CimbReader(const cv::Mat | cv::UMat& img, CimbDecoder& decoder, unsigned color_mode, bool needs_sharpen = false, int color_correction = 2)
	// (cimb_translator/CimbReader.h:16-17). color_mode: the mode's own palette (Config::color_mode(): 1, or 0 in the legacy modes) is the one built.
	template <typename MAT>
	CimbReader(const MAT& img, CimbDecoder& decoder, unsigned color_mode, bool needs_sharpen = false, int color_correction = 2)
	    : CimbReader(img, decoder.decoder(), needs_sharpen, color_correction)
	{
		if (!decoder.good() || color_mode != decoder.decoder().color_mode()) _good = false;
	}

	template <typename MAT>
	CimbReader(const MAT& img, Decoder& decoder, bool needs_sharpen = false, int color_correction = 2)
	{
		detail::with_mat(img, [&](const auto& m) { init(m, decoder, needs_sharpen, color_correction); return 0; });
	}

protected:
	template <typename MAT>
	void init(const MAT& img, Decoder& decoder, bool needs_sharpen, int color_correction)
	{
		_good = false;
		if (!decoder.good()) return;
		std::vector<unsigned char> chunks(decoder.frame_bytes());   // the context's mode's, not mode B's
		uint32_t mask = 0;
		size_t step = (size_t)img.step ? (size_t)img.step : (size_t)img.cols * 3;
		_good = cimbar_hip_decode_frame(decoder.context(), reinterpret_cast<const uint8_t*>(img.data), (unsigned)img.cols, (unsigned)img.rows, step,
		                                needs_sharpen ? 1 : 0, color_correction, chunks.data(), &mask) >= 0;
		if (!_good) return;
		_cells = decoder.total_cells();
		_dim_x = (int)decoder.cells_per_col_x(); _dim_y = (int)decoder.cells_per_col_y(); _offset = (int)decoder.cell_offset();
		_symbols.resize(_cells);
		_colors.resize(_cells);
		_drift.resize((size_t)_cells * 2);
		_good = cimbar_hip_tap(decoder.context(), CIMBAR_HIP_TAP_SYMBOLS, _symbols.data(), _symbols.size()) >= 0 &&
		        cimbar_hip_tap(decoder.context(), CIMBAR_HIP_TAP_COLORS, _colors.data(), _colors.size()) >= 0 &&
		        cimbar_hip_tap(decoder.context(), CIMBAR_HIP_TAP_DRIFT, _drift.data(), _drift.size()) >= 0;
	}

public:
	unsigned read(PositionData& pos)
	{
		if (done()) return 0;
		const unsigned i = _next++;
		int x, y;
		cell_xy(i, x, y);
		pos.i = i;
		pos.x = x + _drift[2 * i];
		pos.y = y + _drift[2 * i + 1];
		return _symbols[i];
	}
	unsigned read_color(const PositionData& pos) const { return pos.i < _colors.size() ? _colors[pos.i] : 0; }
	bool done() const { return !_good || _next >= _cells; }
	unsigned num_reads() const { return _cells; }

	// CellPositions::compute_linear (CellPositions.cpp:5-51) for the decoder's grid: cell pitch 9, 6 marker cells per corner
	void cell_xy(unsigned i, int& x, int& y) const
	{
		const int top_w = _dim_x - 12, top = top_w * 6, mid = _dim_x * (_dim_y - 12);
		if ((int)i < top) { x = _offset + 54 + (int)(i % top_w) * 9; y = _offset + (int)(i / top_w) * 9; }
		else if ((int)i < top + mid) { unsigned j = i - top; x = _offset + (int)(j % _dim_x) * 9; y = _offset + 54 + (int)(j / _dim_x) * 9; }
		else { unsigned j = i - top - mid; x = _offset + 54 + (int)(j % top_w) * 9; y = _offset + (_dim_y - 6) * 9 + (int)(j / top_w) * 9; }
	}

protected:
	bool _good = false;
	unsigned _next = 0, _cells = 0;
	int _dim_x = 112, _dim_y = 112, _offset = 8;
	std::vector<unsigned char> _symbols, _colors;
	std::vector<signed char> _drift;
};

// an owning 8-bit image (what the reference returns as a cv::Mat): shaped like image_view, so it can go straight back into decode()
struct image
{
	std::vector<unsigned char> pixels;
	unsigned char* data = nullptr;
	int cols = 0;
	int rows = 0;
	size_t step = 0;
	int channels = 0;
	image() {}
	image(int w, int h, int c) : pixels((size_t)w * h * c), data(nullptr), cols(w), rows(h), step((size_t)w * c), channels(c) { data = pixels.data(); }
	image(const image& o) : pixels(o.pixels), data(nullptr), cols(o.cols), rows(o.rows), step(o.step), channels(o.channels) { data = pixels.data(); }
	image& operator=(const image& o) { pixels = o.pixels; data = pixels.data(); cols = o.cols; rows = o.rows; step = o.step; channels = o.channels; return *this; }
	bool empty() const { return pixels.empty(); }
	int type() const { return channels; }
	void create(int r, int c, int type_channels) { *this = image(c, r, type_channels); }
};

// Deskewer (Deskewer.h:12-40) and the image preparation of Scanner (Scanner.h:148-165) over the context of a cimbar_amd::Decoder.
// CORNERS is anything with all() returning four points with float-convertible .x / .y in the order top-left, top-right, bottom-left,
// bottom-right -- the reference's Corners (Corners.h:45-53) as is.
class Deskewer
{
public:
	explicit Deskewer(Decoder& decoder) : _dec(decoder) {}

	// Deskewer::deskew(img, corners): an empty image on failure
	template <typename MAT, typename CORNERS>
	image deskew(const MAT& img, const CORNERS& corners)
	{
		return detail::with_mat(img, [&](const auto& m) { return deskew_mat(m, corners); });
	}

	// Scanner::preprocess_image(img, fast = true): 0 / 255 per pixel, what Scanner::scan works on
	template <typename MAT>
	image scan_preprocess(const MAT& img)
	{
		return detail::with_mat(img, [&](const auto& m) { return scan_preprocess_mat(m); });
	}

	template <typename MAT>
	static const unsigned char* dense(const MAT& img, std::vector<unsigned char>& packed) { return dense_rgb(img, packed); }

protected:
	template <typename MAT, typename CORNERS>
	image deskew_mat(const MAT& img, const CORNERS& corners)
	{
		image out;
		if (!_dec.good() || img.cols <= 0 || img.rows <= 0) return out;
		float c8[8];
		const auto pts = corners.all();
		if (pts.size() != 4) return out;
		for (int i = 0; i < 4; ++i) { c8[2 * i] = (float)pts[i].x; c8[2 * i + 1] = (float)pts[i].y; }
		std::vector<unsigned char> packed;
		const unsigned char* src = dense_rgb(img, packed);
		out = image((int)_dec.image_size_x(), (int)_dec.image_size_y(), 3);
		if (cimbar_hip_deskew_batch(_dec.context(), src, (unsigned)img.cols, (unsigned)img.rows, 1, CIMBAR_HIP_MEM_HOST, c8, out.data,
		                            CIMBAR_HIP_MEM_HOST, nullptr) != 0)
			out = image();
		return out;
	}

	template <typename MAT>
	image scan_preprocess_mat(const MAT& img)
	{
		image out;
		if (!_dec.good() || img.cols <= 0 || img.rows <= 0) return out;
		std::vector<unsigned char> packed;
		const unsigned char* src = dense_rgb(img, packed);
		out = image(img.cols, img.rows, 1);
		if (cimbar_hip_scan_preprocess(_dec.context(), src, (unsigned)img.cols, (unsigned)img.rows, 1, CIMBAR_HIP_MEM_HOST, out.data, nullptr,
		                               CIMBAR_HIP_MEM_HOST, nullptr) != 0)
			out = image();
		return out;
	}

	template <typename MAT>
	static const unsigned char* dense_rgb(const MAT& img, std::vector<unsigned char>& packed)
	{
		const unsigned char* src = reinterpret_cast<const unsigned char*>(img.data);
		const size_t dense = (size_t)img.cols * 3, step = (size_t)img.step ? (size_t)img.step : dense;
		if (step == dense) return src;
		packed.resize(dense * (size_t)img.rows);
		for (int y = 0; y < img.rows; ++y)
			for (size_t k = 0; k < dense; ++k) packed[(size_t)y * dense + k] = src[(size_t)y * step + k];
		return packed.data();
	}

	Decoder& _dec;
};

// Extractor (extractor/Extractor.h:11-45): Scanner + Corners + Deskewer, all on the device. MAT is anything with data / cols / rows / step
// that can be re-shaped with create(rows, cols, type) and reports type() -- cv::Mat as is, cimbar_amd::image, or cv::UMat (through getMat);
// `out` may be `img` itself, as in cimbar.cpp:146 (`ext.extract(img, img)`): the capture has been consumed before `out` is re-shaped.
class Extractor
{
public:
	static constexpr int FAILURE = 0;
	static constexpr int SUCCESS = 1;
	static constexpr int NEEDS_SHARPEN = 2;

	// Extractor(padding, image_size, anchor_size): only the defaults the reference's callers use (0, Config's image_size_x x image_size_y, Config's 30)
	explicit Extractor(Decoder& decoder) : _dec(decoder) {}

	template <typename MAT>
	int extract(const MAT& img, MAT& out)
	{
		if (!_dec.good()) return FAILURE;
		const int fw = (int)_dec.image_size_x(), fh = (int)_dec.image_size_y();
		std::vector<unsigned char> frame;
		int type = 0;
		const int status = detail::with_mat(img, [&](const auto& m) {
			if (m.cols <= 0 || m.rows <= 0) return 0;
			type = m.type();
			std::vector<unsigned char> packed;
			const unsigned char* src = Deskewer::dense(m, packed);
			frame.resize((size_t)fw * fh * 3);
			int st = 0;
			if (cimbar_hip_extract_batch(_dec.context(), src, (unsigned)m.cols, (unsigned)m.rows, 1, CIMBAR_HIP_MEM_HOST, frame.data(), &st, _corners,
			                             CIMBAR_HIP_MEM_HOST, nullptr) != 0)
				return 0;
			return st;
		});
		if (status <= 0) return FAILURE;
		out.create(fh, fw, type);
		detail::with_mat_rw(out, [&](auto& m) {
			for (int y = 0; y < fh; ++y)
				for (size_t k = 0; k < (size_t)fw * 3; ++k)
					m.data[(size_t)y * m.step + k] = frame[(size_t)y * fw * 3 + k];
		});
		return status;
	}

	// Corners::all() of the last extract: top-left, top-right, bottom-left, bottom-right (x, y)
	const float* corners() const { return _corners; }

protected:
	Decoder& _dec;
	float _corners[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

}  // namespace cimbar_amd
