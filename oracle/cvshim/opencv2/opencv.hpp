// cv-shim: the *minimum* of the OpenCV API that libcimbar's frame codec touches, so that the
// reference's own sources (Decoder.h, CimbReader.cpp, CimbDecoder.cpp, Encoder.h, CimbWriter.cpp,
// Common.cpp, ...) compile UNMODIFIED from /root/reference into oracle/_ref/libcimbar_ref.so.
//
// TEST INFRASTRUCTURE ONLY. Nothing in the product (libcimbar_amd/) includes this file.
//
// OpenCV itself is not in /root/reference (external system dependency, CMakeLists.txt:38) and is not
// installed in the build container, so every arithmetic rule below is a restatement of OpenCV 4.5.x's
// published behaviour and is marked [assumed-OpenCV]:
//   * cvtColor RGB2GRAY (8u): (R*9798 + G*19235 + B*3735 + (1<<14)) >> 15      (color_rgb.simd.hpp, RGB2Gray<uchar>)
//   * adaptiveThreshold MEAN_C/BINARY: boxFilter(ksize, normalize, BORDER_REPLICATE) to 8u through the
//     ushort column-sum path ((sum + divDelta) * divScale >> 23), then dst = src > mean ? 255 : 0
//     (thresh.cpp adaptiveThreshold, box_filter.simd.hpp ColumnSum<ushort,uchar>)
//   * filter2D 8u->8u with a float kernel: float accumulation, BORDER_REFLECT_101, saturate_cast<uchar>(cvRound)
//   * mean(): per-channel double sum / count
//   * Matx small-matrix ops in the element type, textbook order (matx.hpp MatxMulOp, Matx_FastInvOp<3,3>)
//   * Mat*Mat for CV_32F: double accumulation, cast to float (matmul.simd.hpp GEMMSingleMul<float,double>)
//   * invert(DECOMP_SVD): one-sided Jacobi SVD (lapack.cpp JacobiSVDImpl_<float>) + SVBkSb back-substitution
#pragma once

#include <algorithm>
#include <array>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

typedef unsigned char uchar;

#define CV_VERSION_MAJOR 4
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

// (imgproc.hpp ColorConversionCodes: COLOR_YUV420p2RGB is an alias of COLOR_YUV2RGB_YV12 -- the chroma plane that follows Y is read as V)
enum { COLOR_BGR2RGB = 4, COLOR_RGB2BGR = 4, COLOR_RGBA2RGB = 1, COLOR_RGB2GRAY = 7, COLOR_YUV2RGB_NV12 = 90, COLOR_YUV2RGB_YV12 = 98,
       COLOR_YUV2RGB_IYUV = 100, COLOR_YUV2RGB_I420 = COLOR_YUV2RGB_IYUV, COLOR_YUV420p2RGB = COLOR_YUV2RGB_YV12 };
enum { ADAPTIVE_THRESH_MEAN_C = 0 };
enum { THRESH_BINARY = 0, THRESH_OTSU = 8 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { DECOMP_LU = 0, DECOMP_SVD = 1 };
enum AccessFlag { ACCESS_READ = 1 << 24, ACCESS_RW = 3 << 24, ACCESS_FAST = 1 << 26 };

static inline int cvRound(double v) { return (int)lrint(v); }   // round-half-even under the default FP mode
static inline int cvFloor(double v) { return (int)std::floor(v); }
static inline uchar saturate_u8(int v) { return (uchar)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} bool operator==(const Size& o) const { return width == o.width && height == o.height; } };
struct Point { int x = 0, y = 0; Point() {} Point(int x_, int y_) : x(x_), y(y_) {} };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Rect
{
	int x = 0, y = 0, width = 0, height = 0;
	Rect() {}
	Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
	Rect(const Point& p, const Size& s) : x(p.x), y(p.y), width(s.width), height(s.height) {}
};

struct Scalar
{
	double val[4] = {0, 0, 0, 0};
	Scalar() {}
	Scalar(double a, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
	double& operator[](int i) { return val[i]; }
	const double& operator[](int i) const { return val[i]; }
	bool operator==(const Scalar& o) const { return val[0] == o.val[0] && val[1] == o.val[1] && val[2] == o.val[2] && val[3] == o.val[3]; }
	bool operator!=(const Scalar& o) const { return !(*this == o); }
};

struct Vec3b
{
	uchar val[3] = {0, 0, 0};
	Vec3b() {}
	Vec3b(uchar a, uchar b, uchar c) { val[0] = a; val[1] = b; val[2] = c; }
	Vec3b(std::initializer_list<int> l) { int i = 0; for (int v : l) { if (i < 3) val[i++] = (uchar)v; } }
	uchar& operator[](int i) { return val[i]; }
	const uchar& operator[](int i) const { return val[i]; }
	bool operator==(const Vec3b& o) const { return val[0] == o.val[0] && val[1] == o.val[1] && val[2] == o.val[2]; }
	bool operator!=(const Vec3b& o) const { return !(*this == o); }
};

// ---------------------------------------------------------------- Matx (small fixed matrices, element-type arithmetic)
template <typename T, int M, int N>
struct Matx
{
	T val[M * N];
	Matx() { for (int i = 0; i < M * N; ++i) val[i] = T(0); }
	Matx(T a, T b, T c) { static_assert(M * N == 3, ""); val[0] = a; val[1] = b; val[2] = c; }
	Matx(T a, T b, T c, T d, T e, T f, T g, T h, T i)
	{
		static_assert(M * N == 9, "");
		val[0] = a; val[1] = b; val[2] = c; val[3] = d; val[4] = e; val[5] = f; val[6] = g; val[7] = h; val[8] = i;
	}
	T& operator()(int i, int j) { return val[i * N + j]; }
	const T& operator()(int i, int j) const { return val[i * N + j]; }
	T& operator()(int i) { return val[i]; }
	const T& operator()(int i) const { return val[i]; }

	static Matx diag(const Matx<T, M, 1>& d)
	{
		Matx r;
		for (int i = 0; i < M; ++i) r(i, i) = d.val[i];
		return r;
	}
	Matx div(const Matx& o) const
	{
		Matx r;
		for (int i = 0; i < M * N; ++i) r.val[i] = val[i] / o.val[i];   // matx.hpp: Matx::div -> element-wise a/b
		return r;
	}
	// matx.hpp Matx_FastInvOp<_Tp,3,3>: closed form through the determinant, all in _Tp
	Matx inv(int = DECOMP_LU) const
	{
		static_assert(M == 3 && N == 3, "shim: only 3x3 inv");
		const Matx& a = *this;
		Matx b;
		T d = (T)(a(0,0) * (a(1,1) * a(2,2) - a(2,1) * a(1,2)) - a(0,1) * (a(1,0) * a(2,2) - a(2,0) * a(1,2)) +
		          a(0,2) * (a(1,0) * a(2,1) - a(2,0) * a(1,1)));
		if (d == 0) return b;
		d = 1 / d;
		b(0,0) = (a(1,1) * a(2,2) - a(1,2) * a(2,1)) * d;
		b(0,1) = (a(0,2) * a(2,1) - a(0,1) * a(2,2)) * d;
		b(0,2) = (a(0,1) * a(1,2) - a(0,2) * a(1,1)) * d;
		b(1,0) = (a(1,2) * a(2,0) - a(1,0) * a(2,2)) * d;
		b(1,1) = (a(0,0) * a(2,2) - a(0,2) * a(2,0)) * d;
		b(1,2) = (a(0,2) * a(1,0) - a(0,0) * a(1,2)) * d;
		b(2,0) = (a(1,0) * a(2,1) - a(1,1) * a(2,0)) * d;
		b(2,1) = (a(0,1) * a(2,0) - a(0,0) * a(2,1)) * d;
		b(2,2) = (a(0,0) * a(1,1) - a(0,1) * a(1,0)) * d;
		return b;
	}
};

// matx.hpp MatxMulOp: s = 0; for k: s += a(i,k)*b(k,j), in the element type
template <typename T, int M, int L, int N>
inline Matx<T, M, N> operator*(const Matx<T, M, L>& a, const Matx<T, L, N>& b)
{
	Matx<T, M, N> c;
	for (int i = 0; i < M; ++i)
		for (int j = 0; j < N; ++j)
		{
			T s = 0;
			for (int k = 0; k < L; ++k) s += a(i, k) * b(k, j);
			c(i, j) = s;
		}
	return c;
}

// ---------------------------------------------------------------- Mat
template <typename T> class MatIterator_;
template <typename T> class Mat_;
template <typename T> class MatCommaInitializer_;
class UMat;

class Mat
{
public:
	int rows = 0, cols = 0, dims = 2;
	uchar* data = nullptr;
	size_t step = 0;   // bytes per row

	Mat() {}
	Mat(int r, int c, int type) { create(r, c, type); }
	Mat(int r, int c, int type, const Scalar& s) { create(r, c, type); fill(s); }
	Mat(int r, int c, int type, void* ext, size_t stp = 0)
		: rows(r), cols(c), data((uchar*)ext), _type(type)
	{
		step = stp ? stp : (size_t)c * elemSize();
	}
	template <typename T, int M, int N>
	explicit Mat(const Matx<T, M, N>& m)
	{
		static_assert(sizeof(T) == 4, "shim: float Matx only");
		create(M, N, CV_32F);
		std::memcpy(data, m.val, sizeof(T) * M * N);
	}

	static Mat ones(int r, int c, int type)
	{
		Mat m(r, c, type);
		if (CV_MAT_DEPTH(type) == CV_32F) for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m.ptr<float>(i)[j] = 1.f;
		else m.fill(Scalar(1, 1, 1, 1));
		return m;
	}

	void create(int r, int c, int type)
	{
		if (r == rows && c == cols && type == _type && data && isContinuous()) return;
		rows = r; cols = c; _type = type;
		step = (size_t)c * elemSize();
		_buf = std::shared_ptr<uchar>(new uchar[std::max<size_t>(step * (size_t)r, 1)], std::default_delete<uchar[]>());
		data = _buf.get();
	}

	int type() const { return _type; }
	int depth() const { return CV_MAT_DEPTH(_type); }
	int channels() const { return CV_MAT_CN(_type); }
	size_t elemSize() const { return (size_t)channels() * (depth() == CV_64F ? 8 : (depth() == CV_32F ? 4 : 1)); }
	Size size() const { return Size(cols, rows); }
	bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
	bool isContinuous() const { return step == (size_t)cols * elemSize() || rows <= 1; }
	size_t total() const { return (size_t)rows * cols; }

	template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
	template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
	template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
	template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }

	Mat operator()(const Rect& r) const
	{
		Mat m;
		m.rows = r.height; m.cols = r.width; m._type = _type; m.step = step; m._buf = _buf;
		m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize();
		return m;
	}

	Mat clone() const
	{
		Mat m(rows, cols, _type);
		for (int i = 0; i < rows; ++i) std::memcpy(m.ptr<uchar>(i), ptr<uchar>(i), (size_t)cols * elemSize());
		return m;
	}
	void copyTo(Mat& dst) const
	{
		if (dst.rows != rows || dst.cols != cols || dst._type != _type || !dst.data) dst.create(rows, cols, _type);
		for (int i = 0; i < rows; ++i) std::memmove(dst.ptr<uchar>(i), ptr<uchar>(i), (size_t)cols * elemSize());
	}
	void copyTo(Mat&& dst) const { Mat& d = dst; copyTo(d); }   // ROI temporaries: img.copyTo(canvas(Rect))
	inline UMat getUMat(AccessFlag) const;                        // shares the pixels (cimbar_recv_js.cpp:101: cv::Mat(..).getUMat(ACCESS_RW).clone())

	void push_back(const Mat& row)
	{
		Mat m(rows + row.rows, row.cols, row._type);
		for (int i = 0; i < rows; ++i) std::memcpy(m.ptr<uchar>(i), ptr<uchar>(i), (size_t)cols * elemSize());
		for (int i = 0; i < row.rows; ++i) std::memcpy(m.ptr<uchar>(rows + i), row.ptr<uchar>(i), (size_t)row.cols * row.elemSize());
		*this = m;
	}

	template <typename T> MatIterator_<T> begin();
	template <typename T> MatIterator_<T> end();

	template <typename T, int M, int N>
	operator Matx<T, M, N>() const
	{
		Matx<T, M, N> r;
		for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) r(i, j) = ptr<T>(i)[j];
		return r;
	}

protected:
	void fill(const Scalar& s)
	{
		for (int i = 0; i < rows; ++i)
		{
			uchar* p = ptr<uchar>(i);
			for (int j = 0; j < cols; ++j)
				for (int c = 0; c < channels(); ++c) p[j * channels() + c] = saturate_u8(cvRound(s[c]));
		}
	}
	int _type = 0;
	std::shared_ptr<uchar> _buf;
};

class UMat : public Mat
{
public:
	UMat() {}
	UMat(const Mat& m) : Mat(m) {}
	UMat(int r, int c, int type) : Mat(r, c, type) {}
	Mat getMat(AccessFlag) const { return *this; }
	UMat clone() const { return UMat(Mat::clone()); }
};
inline UMat Mat::getUMat(AccessFlag) const { return UMat(*this); }
inline UMat getUMat(const Mat& m, AccessFlag) { return UMat(m); }   // stands in for Mat::getUMat (cimbar.cpp:132: cv::imread(..).getUMat(ACCESS_RW)); shares the pixels

template <typename T>
class MatIterator_
{
public:
	MatIterator_(Mat* m, size_t pos) : _m(m), _pos(pos) {}
	T& operator*() { return _m->ptr<T>((int)(_pos / _m->cols))[_pos % _m->cols]; }
	MatIterator_& operator++() { ++_pos; return *this; }
	bool operator!=(const MatIterator_& o) const { return _pos != o._pos; }
	bool operator==(const MatIterator_& o) const { return _pos == o._pos; }
protected:
	Mat* _m;
	size_t _pos;
};
template <typename T> inline MatIterator_<T> Mat::begin() { return MatIterator_<T>(this, 0); }
template <typename T> inline MatIterator_<T> Mat::end() { return MatIterator_<T>(this, total()); }

template <typename T>
class Mat_ : public Mat
{
public:
	Mat_() {}
	Mat_(int r, int c) : Mat(r, c, CV_32F) { static_assert(sizeof(T) == 4, "shim: Mat_<float> only"); }
};

template <typename T>
class MatCommaInitializer_
{
public:
	MatCommaInitializer_(const Mat_<T>& m, T first) : _m(m), _idx(0) { put(first); }
	template <typename V> MatCommaInitializer_& operator,(V v) { put((T)v); return *this; }
	operator Mat() const { return _m; }
	operator Mat_<T>() const { return _m; }
protected:
	void put(T v) { _m.template ptr<T>((int)(_idx / _m.cols))[_idx % _m.cols] = v; ++_idx; }
	Mat_<T> _m;
	size_t _idx;
};
template <typename T, typename V>
inline MatCommaInitializer_<T> operator<<(const Mat_<T>& m, V v) { return MatCommaInitializer_<T>(m, (T)v); }

// ---------------------------------------------------------------- what the reference's unit tests use on top of the library code
// [assumed-OpenCV] operator<<(ostream, Matx) = Formatter::FMT_DEFAULT: "[a, b, c;\n d, e, f]" with %.8g for float (%.16g for double)
template <typename T, int M, int N>
inline std::ostream& operator<<(std::ostream& os, const Matx<T, M, N>& m)
{
	os << "[";
	for (int i = 0; i < M; ++i)
	{
		for (int j = 0; j < N; ++j)
		{
			char buf[64];
			std::snprintf(buf, sizeof buf, sizeof(T) == 4 ? "%.8g" : "%.16g", (double)m(i, j));
			os << buf;
			if (j + 1 < N) os << ", ";
		}
		if (i + 1 < M) os << ";\n ";
	}
	return os << "]";
}

// cv::sum: per-channel sum of all elements
inline Scalar sum(const Mat& m)
{
	Scalar s;
	const int cn = m.channels();
	for (int y = 0; y < m.rows; ++y)
	{
		const uchar* p = m.ptr<uchar>(y);
		for (int x = 0; x < m.cols; ++x)
			for (int c = 0; c < cn && c < 4; ++c) s[c] += p[x * cn + c];
	}
	return s;
}
// Mat != Mat (MatExpr compare): 255 where the elements differ, 0 where they are equal, same shape and channels
inline Mat operator!=(const Mat& a, const Mat& b)
{
	Mat r(a.rows, a.cols, a.type());
	const size_t n = (size_t)a.cols * a.elemSize();
	for (int y = 0; y < a.rows; ++y)
	{
		const uchar* pa = a.ptr<uchar>(y); const uchar* pb = b.ptr<uchar>(y); uchar* pr = r.ptr<uchar>(y);
		for (size_t i = 0; i < n; ++i) pr[i] = pa[i] != pb[i] ? 255 : 0;
	}
	return r;
}

// imwrite / imread: the reference's tests write a frame as PNG and read it back -- a LOSSLESS round trip, which is all the shim keeps of it: a private
// container ("CVSHIMG1", rows, cols, channels, the pixels in the caller's channel order), whatever the path's extension says. A file that is not one
// (a real PNG / JPEG: this image has no codec, and the reference's samples/ are not in /root/reference) reads as an empty Mat, like cv::imread of
// a missing file.
inline bool imwrite(const std::string& path, const Mat& m)
{
	std::ofstream f(path, std::ios::binary);
	if (!f) return false;
	const int32_t hdr[3] = {m.rows, m.cols, m.channels()};
	f.write("CVSHIMG1", 8);
	f.write((const char*)hdr, sizeof hdr);
	for (int y = 0; y < m.rows; ++y) f.write((const char*)m.ptr<uchar>(y), (std::streamsize)((size_t)m.cols * m.elemSize()));
	return f.good();
}
inline Mat imread(const std::string& path, int = 1)
{
	std::ifstream f(path, std::ios::binary);
	char magic[8];
	int32_t hdr[3];
	if (!f || !f.read(magic, 8) || std::memcmp(magic, "CVSHIMG1", 8) != 0 || !f.read((char*)hdr, sizeof hdr)) return Mat();
	Mat m(hdr[0], hdr[1], CV_MAKETYPE(CV_8U, hdr[2]));
	f.read((char*)m.data, (std::streamsize)((size_t)hdr[0] * hdr[1] * hdr[2]));
	return f ? m : Mat();
}

// ---------------------------------------------------------------- imgproc subset
// [assumed-OpenCV] color_rgb.simd.hpp RGB2Gray<uchar>: 15-bit fixed point, R2Y=9798 G2Y=19235 B2Y=3735
inline void cvtColor(const Mat& src_, Mat& dst, int code)
{
	Mat src = src_;   // keeps the buffer alive for in-place calls
	if (code == COLOR_RGB2GRAY)
	{
		Mat out(src.rows, src.cols, CV_8UC1);
		int cn = src.channels();
		for (int y = 0; y < src.rows; ++y)
		{
			const uchar* s = src.ptr<uchar>(y);
			uchar* d = out.ptr<uchar>(y);
			for (int x = 0; x < src.cols; ++x, s += cn)
				d[x] = (uchar)((s[0] * 9798 + s[1] * 19235 + s[2] * 3735 + (1 << 14)) >> 15);
		}
		dst = out;
	}
	else if (code == COLOR_RGBA2RGB)
	{
		Mat out(src.rows, src.cols, CV_8UC3);
		for (int y = 0; y < src.rows; ++y)
		{
			const uchar* s = src.ptr<uchar>(y);
			uchar* d = out.ptr<uchar>(y);
			for (int x = 0; x < src.cols; ++x) { d[3*x] = s[4*x]; d[3*x+1] = s[4*x+1]; d[3*x+2] = s[4*x+2]; }
		}
		dst = out;
	}
	else if (code == COLOR_BGR2RGB)
	{
		Mat out(src.rows, src.cols, CV_8UC3);
		for (int y = 0; y < src.rows; ++y)
		{
			const uchar* s = src.ptr<uchar>(y);
			uchar* d = out.ptr<uchar>(y);
			for (int x = 0; x < src.cols; ++x) { d[3*x] = s[3*x+2]; d[3*x+1] = s[3*x+1]; d[3*x+2] = s[3*x]; }
		}
		dst = out;
	}
	else if (code == COLOR_YUV2RGB_NV12 || code == COLOR_YUV2RGB_YV12 || code == COLOR_YUV2RGB_IYUV)
	{
		// [assumed-OpenCV] color_yuv.simd.hpp, YUV 4:2:0 -> RGB (cvtColorTwoPlaneYUV2BGR / cvtColorThreePlaneYUV2BGR): the source is a one-channel
		// Mat of height*3/2 rows (color.cpp asserts width % 2 == 0 && rows % 3 == 0); ITU-R BT.601 in 20-bit fixed point,
		//   ruv = 2^19 + CVR*(v-128), guv = 2^19 + CVG*(v-128) + CUG*(u-128), buv = 2^19 + CUB*(u-128), y = max(0, Y-16)*CY,
		//   c = saturate_cast<uchar>((y + cuv) >> 20)
		// NV12: rows of interleaved (U, V) pairs behind the Y plane (uIdx 0). YV12 (= COLOR_YUV420p2RGB): a (w/2) x (h/2) V plane, then U;
		// IYUV / I420: U, then V. The chroma planes are addressed the way YUV420p2RGB8Invoker does it (two chroma rows per Mat row of `step`
		// bytes, the odd-quarter-height offset of cvtColorThreePlaneYUV2BGR), which for a continuous Mat is simply plane-contiguous.
		if (src.type() != CV_8UC1 || src.cols % 2 != 0 || src.rows % 3 != 0) { std::cerr << "cv-shim: cvtColor YUV420: CV_8UC1, even width, rows % 3 == 0" << std::endl; std::abort(); }
		const int W = src.cols, H = src.rows * 2 / 3;
		const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;
		Mat out(H, W, CV_8UC3);
		const size_t stride = src.step;
		const uchar* y0 = src.ptr<uchar>(0);
		const uchar* p1 = y0 + stride * (size_t)H;                                                            // first chroma plane (or the UV rows)
		const uchar* p2 = y0 + stride * (size_t)(H + H / 4) + (size_t)(W / 2) * ((H % 4) / 2);                 // second chroma plane
		int step1 = 0, step2 = H % 4 == 2 ? 1 : 0;
		const bool planar = code != COLOR_YUV2RGB_NV12;
		const uchar* up = p1; const uchar* vp = p2; int us = step1, vs = step2;
		if (code == COLOR_YUV2RGB_YV12) { std::swap(up, vp); std::swap(us, vs); }
		const int uvsteps[2] = {W / 2, (int)stride - W / 2};
		for (int j = 0; j < H; j += 2)
		{
			const uchar* ya = y0 + stride * (size_t)j;
			const uchar* yb = ya + stride;
			uchar* ra = out.ptr<uchar>(j);
			uchar* rb = j + 1 < H ? out.ptr<uchar>(j + 1) : nullptr;
			const uchar* uvrow = p1 + stride * (size_t)(j / 2);
			for (int i = 0; i < W / 2; ++i)
			{
				const int u = planar ? up[i] : uvrow[2 * i], v = planar ? vp[i] : uvrow[2 * i + 1];
				const int uu = u - 128, vv = v - 128;
				const int ruv = (1 << (SHIFT - 1)) + CVR * vv, guv = (1 << (SHIFT - 1)) + CVG * vv + CUG * uu, buv = (1 << (SHIFT - 1)) + CUB * uu;
				auto put = [&](uchar* d, int Y) {
					const int y = std::max(0, Y - 16) * CY;
					d[0] = saturate_u8((y + ruv) >> SHIFT); d[1] = saturate_u8((y + guv) >> SHIFT); d[2] = saturate_u8((y + buv) >> SHIFT);
				};
				put(ra + 6 * i, ya[2 * i]); put(ra + 6 * i + 3, ya[2 * i + 1]);
				if (rb) { put(rb + 6 * i, yb[2 * i]); put(rb + 6 * i + 3, yb[2 * i + 1]); }
			}
			if (planar) { up += uvsteps[(us++) & 1]; vp += uvsteps[(vs++) & 1]; }
		}
		dst = out;
	}
	else
	{
		std::cerr << "cv-shim: unsupported cvtColor code " << code << std::endl;
		std::abort();
	}
}

// [assumed-OpenCV] resize.cpp, INTER_LINEAR for CV_8U (resizeGeneric_ + HResizeLinear / VResizeLinear<uchar, int, short, FixedPtCast<.., 22>>):
//   fx = (dx + 0.5) * (src.w / dst.w) - 0.5 (float); sx = floor(fx), fx -= sx; sx < 0 -> (0, fx 0); sx >= src.w - 1 -> (src.w - 1, fx 0)
//   weights as shorts: saturate_cast<short>((1 - fx) * 2048), saturate_cast<short>(fx * 2048); horizontal pass in int, vertical pass
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2. (The exact-2x shrink, which OpenCV turns into INTER_AREA, is not taken by any caller.)
inline void resize(const Mat& src_, Mat& dst, Size dsize, double = 0, double = 0, int = INTER_LINEAR)
{
	Mat src = src_;
	const int cn = src.channels(), sw = src.cols, sh = src.rows, dw = dsize.width, dh = dsize.height;
	if (src.depth() != CV_8U || (sw == 2 * dw && sh == 2 * dh)) { std::cerr << "cv-shim: resize: CV_8U, INTER_LINEAR, not the exact 2x shrink" << std::endl; std::abort(); }
	auto sat_short = [](float v) { int i = cvRound(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); };
	std::vector<int> xofs(dw), yofs(dh);
	std::vector<short> xa(2 * dw), ya(2 * dh);
	const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
	for (int dx = 0; dx < dw; ++dx)
	{
		float fx = (float)((dx + 0.5) * scale_x - 0.5);
		int sx = cvFloor(fx);
		fx -= sx;
		if (sx < 0) { fx = 0; sx = 0; }
		if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
		xofs[dx] = sx; xa[2 * dx] = sat_short((1.f - fx) * 2048); xa[2 * dx + 1] = sat_short(fx * 2048);
	}
	for (int dy = 0; dy < dh; ++dy)
	{
		float fy = (float)((dy + 0.5) * scale_y - 0.5);
		int sy = cvFloor(fy);
		fy -= sy;
		yofs[dy] = sy; ya[2 * dy] = sat_short((1.f - fy) * 2048); ya[2 * dy + 1] = sat_short(fy * 2048);
	}
	Mat out(dh, dw, src.type());
	std::vector<int> r0((size_t)dw * cn), r1((size_t)dw * cn);
	auto hrow = [&](int sy, std::vector<int>& row) {
		sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);                     // (the vertical taps clip to the image: resizeGeneric_Invoker's clip())
		const uchar* S = src.ptr<uchar>(sy);
		for (int dx = 0; dx < dw; ++dx)
		{
			const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
			for (int c = 0; c < cn; ++c) row[(size_t)dx * cn + c] = S[sx * cn + c] * xa[2 * dx] + S[sx1 * cn + c] * xa[2 * dx + 1];
		}
	};
	for (int dy = 0; dy < dh; ++dy)
	{
		hrow(yofs[dy], r0); hrow(yofs[dy] + 1, r1);
		uchar* D = out.ptr<uchar>(dy);
		const int b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
		for (size_t i = 0; i < (size_t)dw * cn; ++i) D[i] = (uchar)((((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2);
	}
	dst = out;
}

// [assumed-OpenCV] filter2D, ddepth=-1, 8u, float kernel, anchor centre, delta 0, BORDER_REFLECT_101;
// accumulation in float over the non-zero taps, then saturate_cast<uchar>(float) = clamp(cvRound(v)).
inline void filter2D(const Mat& src_, Mat& dst, int, const Mat& kernel)
{
	Mat src = src_.clone();
	Mat out(src.rows, src.cols, src.type());
	int kh = kernel.rows, kw = kernel.cols, ay = kh / 2, ax = kw / 2;
	auto refl = [](int p, int n) { if (p < 0) p = -p; if (p >= n) p = 2 * n - 2 - p; return p; };
	for (int y = 0; y < src.rows; ++y)
		for (int x = 0; x < src.cols; ++x)
		{
			float acc = 0.f;
			for (int i = 0; i < kh; ++i)
				for (int j = 0; j < kw; ++j)
				{
					float k = kernel.ptr<float>(i)[j];
					if (k == 0.f) continue;
					acc += k * (float)src.ptr<uchar>(refl(y + i - ay, src.rows))[refl(x + j - ax, src.cols)];
				}
			out.ptr<uchar>(y)[x] = saturate_u8(cvRound(acc));
		}
	dst = out;
}

// [assumed-OpenCV] thresh.cpp adaptiveThreshold(MEAN_C, THRESH_BINARY): mean = boxFilter 8u->8u
// (BORDER_REPLICATE), ColumnSum<ushort,uchar> normalisation: d=ksize^2, scalef=(1<<23)/d,
// divScale=floor(scalef), divDelta=d/2 (+1 if frac<0.5, else divScale+1); dst = (src - mean > -ceil(C)) ? maxval : 0
inline void adaptiveThreshold(const Mat& src_, Mat& dst, double maxval, int, int, int blockSize, double C)
{
	Mat src = src_.clone();
	Mat out(src.rows, src.cols, CV_8UC1);
	const int r = blockSize / 2, d = blockSize * blockSize, SHIFT = 23;
	double scalef = ((double)(1 << SHIFT)) / d;
	int divScale = cvFloor(scalef);
	scalef -= divScale;
	int divDelta = d / 2;
	if (scalef < 0.5) divDelta++; else divScale++;
	const int idelta = (int)std::ceil(C);
	const uchar on = saturate_u8(cvRound(maxval));
	const int W = src.cols, H = src.rows;
	// separable box sum with running windows (what OpenCV's RowSum/ColumnSum do), BORDER_REPLICATE on both axes
	std::vector<uint16_t> rowsum((size_t)W * H);
	for (int y = 0; y < H; ++y)
	{
		const uchar* s = src.ptr<uchar>(y);
		uint16_t* o = rowsum.data() + (size_t)y * W;
		int acc = 0;
		for (int k = -r; k <= r; ++k) acc += s[k < 0 ? 0 : (k >= W ? W - 1 : k)];
		o[0] = (uint16_t)acc;
		for (int x = 1; x < W; ++x)
		{
			int xin = x + r, xout = x - r - 1;
			acc += s[xin >= W ? W - 1 : xin] - s[xout < 0 ? 0 : xout];
			o[x] = (uint16_t)acc;
		}
	}
	std::vector<int> col((size_t)W, 0);
	for (int k = -r; k <= r; ++k)
	{
		const uint16_t* rs = rowsum.data() + (size_t)(k < 0 ? 0 : (k >= H ? H - 1 : k)) * W;
		for (int x = 0; x < W; ++x) col[x] += rs[x];
	}
	for (int y = 0; y < H; ++y)
	{
		if (y > 0)
		{
			int yin = y + r, yout = y - r - 1;
			const uint16_t* rin = rowsum.data() + (size_t)(yin >= H ? H - 1 : yin) * W;
			const uint16_t* rout = rowsum.data() + (size_t)(yout < 0 ? 0 : yout) * W;
			for (int x = 0; x < W; ++x) col[x] += rin[x] - rout[x];
		}
		const uchar* s = src.ptr<uchar>(y);
		uchar* o = out.ptr<uchar>(y);
		for (int x = 0; x < W; ++x)
		{
			int mean = (int)(((unsigned)(col[x] + divDelta) * (unsigned)divScale) >> SHIFT);
			o[x] = ((int)s[x] - mean > -idelta) ? on : 0;
		}
	}
	dst = out;
}

// [assumed-OpenCV] mean(): double sum per channel / pixel count
inline Scalar mean(const Mat& m)
{
	Scalar s;
	int cn = m.channels();
	for (int y = 0; y < m.rows; ++y)
	{
		const uchar* p = m.ptr<uchar>(y);
		for (int x = 0; x < m.cols; ++x)
			for (int c = 0; c < cn && c < 4; ++c) s[c] += p[x * cn + c];
	}
	double n = (double)m.total();
	if (n > 0) for (int c = 0; c < 4; ++c) s[c] /= n;
	return s;
}

inline void transpose(const Mat& src_, Mat& dst)
{
	Mat src = src_;
	Mat out(src.cols, src.rows, src.type());
	for (int i = 0; i < src.rows; ++i)
		for (int j = 0; j < src.cols; ++j) out.ptr<float>(j)[i] = src.ptr<float>(i)[j];
	dst = out;
}

// [assumed-OpenCV] GEMMSingleMul<float,double>: double accumulator over k ascending, cast to float
inline Mat operator*(const Mat& a, const Mat& b)
{
	Mat c(a.rows, b.cols, CV_32F);
	for (int i = 0; i < a.rows; ++i)
		for (int j = 0; j < b.cols; ++j)
		{
			double s = 0;
			for (int k = 0; k < a.cols; ++k) s += (double)a.ptr<float>(i)[k] * (double)b.ptr<float>(k)[j];
			c.ptr<float>(i)[j] = (float)s;
		}
	return c;
}

namespace shim_detail {
// [assumed-OpenCV] lapack.cpp JacobiSVDImpl_<float>: one-sided (Hestenes) Jacobi on the n rows (length m) of At.
// On return the rows of At are the (normalised) left factors, W the singular values (descending), Vt the
// accumulated rotations. hypot() is spelled sqrt(p*p+beta*beta) so that CPU and GPU restatements agree bit for bit.
inline void jacobi_svd_f32(float* At, int astep, float* Wout, float* Vt, int vstep, int m, int n, int n1)
{
	const double minval = FLT_MIN;
	const float eps = FLT_EPSILON * 2;
	std::vector<double> W(n);
	int max_iter = std::max(m, 30);
	for (int i = 0; i < n; ++i)
	{
		double sd = 0;
		for (int k = 0; k < m; ++k) { float t = At[i * astep + k]; sd += (double)t * t; }
		W[i] = sd;
		for (int k = 0; k < n; ++k) Vt[i * vstep + k] = 0;
		Vt[i * vstep + i] = 1;
	}
	for (int iter = 0; iter < max_iter; ++iter)
	{
		bool changed = false;
		for (int i = 0; i < n - 1; ++i)
			for (int j = i + 1; j < n; ++j)
			{
				float* Ai = At + i * astep; float* Aj = At + j * astep;
				double a = W[i], p = 0, b = W[j];
				for (int k = 0; k < m; ++k) p += (double)Ai[k] * Aj[k];
				if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
				p *= 2;
				double beta = a - b, gamma = std::sqrt(p * p + beta * beta);
				float c, s;
				if (beta < 0)
				{
					double delta = (gamma - beta) * 0.5;
					s = (float)std::sqrt(delta / gamma);
					c = (float)(p / (gamma * s * 2));
				}
				else
				{
					c = (float)std::sqrt((gamma + beta) / (gamma * 2));
					s = (float)(p / (gamma * c * 2));
				}
				a = b = 0;
				for (int k = 0; k < m; ++k)
				{
					float t0 = c * Ai[k] + s * Aj[k];
					float t1 = -s * Ai[k] + c * Aj[k];
					Ai[k] = t0; Aj[k] = t1;
					a += (double)t0 * t0; b += (double)t1 * t1;
				}
				W[i] = a; W[j] = b;
				changed = true;
				float* Vi = Vt + i * vstep; float* Vj = Vt + j * vstep;
				for (int k = 0; k < n; ++k)
				{
					float t0 = c * Vi[k] + s * Vj[k];
					float t1 = -s * Vi[k] + c * Vj[k];
					Vi[k] = t0; Vj[k] = t1;
				}
			}
		if (!changed) break;
	}
	for (int i = 0; i < n; ++i)
	{
		double sd = 0;
		for (int k = 0; k < m; ++k) { float t = At[i * astep + k]; sd += (double)t * t; }
		W[i] = std::sqrt(sd);
	}
	for (int i = 0; i < n - 1; ++i)
	{
		int j = i;
		for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
		if (i != j)
		{
			std::swap(W[i], W[j]);
			for (int k = 0; k < m; ++k) std::swap(At[i * astep + k], At[j * astep + k]);
			for (int k = 0; k < n; ++k) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
		}
	}
	for (int i = 0; i < n; ++i) Wout[i] = (float)W[i];
	for (int i = 0; i < n1; ++i)
	{
		double sd = i < n ? W[i] : 0;
		// (OpenCV regenerates a random orthogonal vector for zero singular values; a rank-deficient colour
		//  sample set never reaches the classifier with a usable matrix anyway -- left as zero here.)
		float s = (float)(sd > minval ? 1 / sd : 0.);
		for (int k = 0; k < m; ++k) At[i * astep + k] *= s;
	}
}
}  // namespace shim_detail

// [assumed-OpenCV] invert(src, dst, DECOMP_SVD) for CV_32F, any shape: SVD::compute + SVD::backSubst(identity)
inline double invert(const Mat& src_, Mat& dst, int method)
{
	if (method != DECOMP_SVD || src_.depth() != CV_32F) { std::cerr << "cv-shim: invert: only DECOMP_SVD/CV_32F" << std::endl; std::abort(); }
	Mat src = src_;
	int m = src.rows, n = src.cols;
	bool at = false;
	if (m < n) { std::swap(m, n); at = true; }
	// temp_a: n rows of length m (== src if at, else src^T)
	std::vector<float> A((size_t)n * m), V((size_t)n * n), W(n);
	for (int i = 0; i < n; ++i)
		for (int k = 0; k < m; ++k) A[(size_t)i * m + k] = at ? src.ptr<float>(i)[k] : src.ptr<float>(k)[i];
	shim_detail::jacobi_svd_f32(A.data(), m, W.data(), V.data(), n, m, n, n);
	// at:  u = V^T (src.rows x nm), vt = A (nm x src.cols);  !at: u = A^T, vt = V
	int M = src.rows, N = src.cols, nm = std::min(M, N);
	auto U = [&](int r, int k) { return at ? V[(size_t)k * n + r] : A[(size_t)k * m + r]; };    // u(r,k), r<M
	auto VT = [&](int k, int c) { return at ? A[(size_t)k * m + c] : V[(size_t)k * n + c]; };  // vt(k,c), c<N
	// lapack.cpp SVBkSbImpl_ with b == identity: x (N x M) += v_k (x) (u_k / w_k), skipping w_k <= eps*sum(w)
	Mat out(N, M, CV_32F);
	for (int i = 0; i < N; ++i) for (int j = 0; j < M; ++j) out.ptr<float>(i)[j] = 0;
	double threshold = 0;
	for (int i = 0; i < nm; ++i) threshold += W[i];
	threshold *= (double)(FLT_EPSILON * 2);
	std::vector<double> buffer(M);
	for (int k = 0; k < nm; ++k)
	{
		double wi = W[k];
		if (std::abs(wi) <= threshold) continue;
		wi = 1 / wi;
		for (int j = 0; j < M; ++j) buffer[j] = U(j, k) * wi;
		for (int i = 0; i < N; ++i)
		{
			float s = VT(k, i);
			float* y = out.ptr<float>(i);
			for (int j = 0; j < M; ++j) y[j] = (float)(y[j] + s * buffer[j]);
		}
	}
	dst = out;
	return W[0] >= FLT_EPSILON ? W[n - 1] / W[0] : 0;
}

// ---------------------------------------------------------------------------------------------- extractor (Scanner / Deskewer)
// [assumed-OpenCV] smooth.dispatch.cpp GaussianBlur for CV_8U, sigma <= 0, ksize 3 | 5: the small fixed kernels [1 2 1]/4 and
// [1 4 6 4 1]/16 (getGaussianKernel's small_gaussian_tab), run by the fixed-point path (ufixedpoint16 8.8 weights, horizontal
// pass in 16 bit, vertical pass in 32 bit, one final rounding): dst = (sum_ij w_i w_j p + 2^(s-1)) >> s with integer weights
// summing to 2^s (s = 4 | 8). BORDER_DEFAULT = BORDER_REFLECT_101. ksize 9: getGaussianKernel(9, sigma <= 0) -> sigma = 0.3*((9-1)*0.5 - 1) + 0.8 = 1.7, exp(-x^2 / (2 sigma^2)) normalised, then the 8.8 fixed-point
// weights of getGaussianKernelFixedPoint_ED (round outside-in with the error carried, centre = 256 - the rest): 256 * k = 3.80 12.75 30.29 50.90
// 60.51 -> {4, 13, 30, 51, 60, 51, 30, 13, 4} (plain rounding gives the same, no value is near a half), s = 16. ksize 17 (captures of 4500 px and
// more on the short side): sigma = 2.9, 256 * k = 0.786 1.919 4.156 7.992 13.647 20.691 27.853 33.292 35.331 -> with the error carried
// outside-in {1, 2, 4, 8, 13, 21, 28, 33, 36, ...} (the fifth weight is 13.4993 before rounding: 0.0007 below the boundary), s = 16.
inline void GaussianBlur(const Mat& src_, Mat& dst, Size ksize, double, double = 0)
{
	if (src_.type() != CV_8UC1 || ksize.width != ksize.height || (ksize.width != 3 && ksize.width != 5 && ksize.width != 9 && ksize.width != 17))
	{ std::cerr << "cv-shim: GaussianBlur: CV_8UC1 with ksize 3, 5, 9 or 17 only" << std::endl; std::abort(); }
	Mat src = src_.clone();
	const int W = src.cols, H = src.rows, r = ksize.width / 2;
	static const int k3[3] = {1, 2, 1}, k5[5] = {1, 4, 6, 4, 1}, k9[9] = {4, 13, 30, 51, 60, 51, 30, 13, 4},
	                 k17[17] = {1, 2, 4, 8, 13, 21, 28, 33, 36, 33, 28, 21, 13, 8, 4, 2, 1};
	const int* k = r == 1 ? k3 : (r == 2 ? k5 : (r == 4 ? k9 : k17));
	const int shift = r == 1 ? 4 : (r == 2 ? 8 : 16);
	auto refl = [](int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; };
	std::vector<int> h((size_t)W * H);
	for (int y = 0; y < H; ++y)
	{
		const uchar* s = src.ptr<uchar>(y);
		for (int x = 0; x < W; ++x)
		{
			int acc = 0;
			for (int t = -r; t <= r; ++t) acc += k[t + r] * s[refl(x + t, W)];
			h[(size_t)y * W + x] = acc;
		}
	}
	Mat out(H, W, CV_8UC1);
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x)
		{
			int acc = 0;
			for (int t = -r; t <= r; ++t) acc += k[t + r] * h[(size_t)refl(y + t, H) * W + x];
			out.ptr<uchar>(y)[x] = (uchar)((acc + (1 << (shift - 1))) >> shift);
		}
	dst = out;
}

// [assumed-OpenCV] thresh.cpp threshold(THRESH_BINARY | THRESH_OTSU) for CV_8UC1: getThreshVal_Otsu_8u, then dst = src > t ? maxval : 0
inline double threshold(const Mat& src_, Mat& dst, double thresh, double maxval, int type)
{
	if (src_.type() != CV_8UC1) { std::cerr << "cv-shim: threshold: CV_8UC1 only" << std::endl; std::abort(); }
	Mat src = src_.clone();
	const int W = src.cols, H = src.rows;
	if (type & THRESH_OTSU)
	{
		int hist[256] = {0};
		for (int y = 0; y < H; ++y) { const uchar* s = src.ptr<uchar>(y); for (int x = 0; x < W; ++x) hist[s[x]]++; }
		double mu = 0, scale = 1. / ((double)W * H);
		for (int i = 0; i < 256; ++i) mu += i * (double)hist[i];
		mu *= scale;
		double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
		for (int i = 0; i < 256; ++i)
		{
			double p_i, q2, mu2, sigma;
			p_i = hist[i] * scale;
			mu1 *= q1;
			q1 += p_i;
			q2 = 1. - q1;
			if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
			mu1 = (mu1 + i * p_i) / q1;
			mu2 = (mu - q1 * mu1) / q2;
			sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
			if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
		}
		thresh = max_val;
	}
	const int t = cvFloor(thresh);
	const uchar on = saturate_u8(cvRound(maxval));
	Mat out(H, W, CV_8UC1);
	for (int y = 0; y < H; ++y) { const uchar* s = src.ptr<uchar>(y); uchar* o = out.ptr<uchar>(y); for (int x = 0; x < W; ++x) o[x] = s[x] > t ? on : 0; }
	dst = out;
	return thresh;
}

namespace shim_detail {
// [assumed-OpenCV] hal LU (matrix_decomp.cpp LUImpl<double>): partial pivoting, eliminate with d = -1/pivot, back-substitute
inline int lu_solve(double* A, int m, double* b)
{
	const double eps = DBL_EPSILON * 100;
	for (int i = 0; i < m; ++i)
	{
		int k = i;
		for (int j = i + 1; j < m; ++j) if (std::abs(A[j * m + i]) > std::abs(A[k * m + i])) k = j;
		if (std::abs(A[k * m + i]) < eps) return 0;
		if (k != i) { for (int j = i; j < m; ++j) std::swap(A[i * m + j], A[k * m + j]); std::swap(b[i], b[k]); }
		double d = -1 / A[i * m + i];
		for (int j = i + 1; j < m; ++j)
		{
			double alpha = A[j * m + i] * d;
			for (int c = i + 1; c < m; ++c) A[j * m + c] += alpha * A[i * m + c];
			b[j] += alpha * b[i];
		}
	}
	for (int i = m - 1; i >= 0; --i)
	{
		double s = b[i];
		for (int c = i + 1; c < m; ++c) s -= A[i * m + c] * b[c];
		b[i] = s / A[i * m + i];
	}
	return 1;
}
// [assumed-OpenCV] lapack.cpp invert() closed form for 3x3 CV_64F (det3, then the adjugate times 1/det)
inline bool invert3x3(const double* S, double* t)
{
	double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
	if (d == 0.) return false;
	d = 1. / d;
	t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
	t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
	t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
	return true;
}
}  // namespace shim_detail

// [assumed-OpenCV] imgwarp.cpp getPerspectiveTransform(src[4], dst[4]): 8x8 system, solve(DECOMP_LU), M[8] = 1
inline Mat getPerspectiveTransform(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, int = DECOMP_LU)
{
	double a[8][8], b[8];
	for (int i = 0; i < 4; ++i)
	{
		a[i][0] = a[i + 4][3] = src[i].x;
		a[i][1] = a[i + 4][4] = src[i].y;
		a[i][2] = a[i + 4][5] = 1;
		a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
		a[i][6] = -src[i].x * dst[i].x;
		a[i][7] = -src[i].y * dst[i].x;
		a[i + 4][6] = -src[i].x * dst[i].y;
		a[i + 4][7] = -src[i].y * dst[i].y;
		b[i] = dst[i].x;
		b[i + 4] = dst[i].y;
	}
	Mat M(3, 3, CV_64FC1);
	double* m = M.ptr<double>(0);
	if (!shim_detail::lu_solve(&a[0][0], 8, b)) for (int i = 0; i < 8; ++i) b[i] = 0;
	for (int i = 0; i < 8; ++i) m[i] = b[i];
	m[8] = 1.;
	return M;
}

// [assumed-OpenCV] imgwarp.cpp warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) for CV_8UC3: M inverted (closed form), then per 64x16
// block (BLOCK_SZ 32 -> bh0 = 16, bw0 = 64) the source coordinate of every destination pixel in 1/32-pixel fixed point,
//   X0 = M0*x + M1*(y+y1) + M2 at the block's left edge, fX = (X0 + M0*x1) * (32 / (W0 + M6*x1)), X = cvRound(fX),
// and remap's fixed-point bilinear: weights (32-fx)(32-fy)*32 .. (sum 2^15), out = (sum w*p + 2^14) >> 15, taps outside the source = 0
inline void warpPerspective(const Mat& src_, Mat& dst, const Mat& M0, Size dsize, int flags = INTER_LINEAR)
{
	if (src_.type() != CV_8UC3 || M0.type() != CV_64FC1 || (flags & 7) != INTER_LINEAR)
	{ std::cerr << "cv-shim: warpPerspective: CV_8UC3, CV_64F matrix, INTER_LINEAR only" << std::endl; std::abort(); }
	Mat src = src_.clone();
	double Min[9], M[9];
	for (int i = 0; i < 9; ++i) Min[i] = M0.ptr<double>(i / 3)[i % 3];
	if (!shim_detail::invert3x3(Min, M)) for (int i = 0; i < 9; ++i) M[i] = 0;
	const int width = dsize.width, height = dsize.height, sw = src.cols, sh = src.rows;
	Mat out(height, width, CV_8UC3);
	const int BLOCK_SZ = 32;
	int bh0 = std::min(BLOCK_SZ / 2, height), bw0 = std::min(BLOCK_SZ * BLOCK_SZ / bh0, width);
	bh0 = std::min(BLOCK_SZ * BLOCK_SZ / bw0, height);
	for (int y = 0; y < height; y += bh0)
		for (int x = 0; x < width; x += bw0)
		{
			const int bw = std::min(bw0, width - x), bh = std::min(bh0, height - y);
			for (int y1 = 0; y1 < bh; ++y1)
			{
				const double X0 = M[0] * x + M[1] * (y + y1) + M[2];
				const double Y0 = M[3] * x + M[4] * (y + y1) + M[5];
				const double W0 = M[6] * x + M[7] * (y + y1) + M[8];
				uchar* o = out.ptr<uchar>(y + y1) + (size_t)x * 3;
				for (int x1 = 0; x1 < bw; ++x1)
				{
					double W = W0 + M[6] * x1;
					W = W ? 32. / W : 0;
					const double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, (X0 + M[0] * x1) * W));
					const double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, (Y0 + M[3] * x1) * W));
					const int X = cvRound(fX), Y = cvRound(fY);
					auto sat16 = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
					const int sx = sat16(X >> 5), sy = sat16(Y >> 5), fx = X & 31, fy = Y & 31;
					const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
					for (int c = 0; c < 3; ++c)
					{
						auto px = [&](int yy, int xx) -> int { return (xx < 0 || xx >= sw || yy < 0 || yy >= sh) ? 0 : src.ptr<uchar>(yy)[xx * 3 + c]; };
						const int v = px(sy, sx) * w00 + px(sy, sx + 1) * w01 + px(sy + 1, sx) * w10 + px(sy + 1, sx + 1) * w11;
						o[x1 * 3 + c] = (uchar)((v + (1 << 14)) >> 15);
					}
				}
			}
		}
	dst = out;
}

}  // namespace cv
