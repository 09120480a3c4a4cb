// Streaming-read probe for K1's access pattern (not part of the product): how fast can 1024 frames x 3 MiB be pulled
// through registers with (a) K1's 48-byte-per-lane row ownership, (b) fully coalesced 16-byte lanes, (c) a flat grid-stride read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr size_t FRAME = 1024ull * 1024 * 3;

#ifndef WPE
#define WPE 8
#endif
template <int MODE, int DEPTH, int ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_stream(const uint8_t* __restrict__ rgb, uint32_t* __restrict__ out, int strips)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int strip = blockIdx.x * 4 + wave;
	if (strip >= strips) return;
	const uint8_t* frame = rgb + (size_t)blockIdx.y * FRAME;
	const int y0 = strip * ROWS;
	uint4 buf[DEPTH][3];
	auto ld = [&](int y, uint4 (&b)[3]) {
		const uint8_t* row = frame + (size_t)y * 3072;
		if (MODE == 0) { const uint4* p = (const uint4*)(row + lane * 48); b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; }
		else { const uint4* p = (const uint4*)(row + lane * 16); b[0] = p[0]; b[1] = p[64]; b[2] = p[128]; }
	};
#pragma unroll
	for (int k = 0; k < DEPTH; ++k) ld(y0 + k, buf[k]);
	uint32_t acc = 0;
	for (int t0 = 0; t0 < ROWS; t0 += DEPTH) {
#pragma unroll
		for (int s = 0; s < DEPTH; ++s) {
			const int t = t0 + s;
			if (t < ROWS) {
				uint4 (&b)[3] = buf[s];
				acc ^= b[0].x ^ b[0].y ^ b[0].z ^ b[0].w ^ b[1].x ^ b[1].y ^ b[1].z ^ b[1].w ^ b[2].x ^ b[2].y ^ b[2].z ^ b[2].w;
				if (t + DEPTH < ROWS) ld(y0 + t + DEPTH, b);
			}
		}
	}
	if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_flat(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
	uint32_t acc = 0;
	for (; i + 3 * stride < n; i += 4 * stride) {
		uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
		acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
	}
	for (; i < n; i += stride) { uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
	if (acc == 0x12345678u) out[0] = acc;
}

template <typename F> float timeit(F f, int reps = 10)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	f(); hipDeviceSynchronize();
	hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main()
{
	const int F = 1024;
	uint8_t* d; uint32_t* o;
	CK(hipMalloc(&d, F * FRAME)); CK(hipMalloc(&o, 64));
	CK(hipMemset(d, 0x5a, F * FRAME));
	// randomise a bit so the DVFS does not see an all-constant buffer
	std::vector<uint32_t> h(1 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
	for (size_t off = 0; off < F * FRAME; off += h.size() * 4) CK(hipMemcpy(d + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	const double gb = F * (double)FRAME / 1e9;
#define RUN(MODE, DEPTH, ROWS, label) { float ms = timeit([&] { hipLaunchKernelGGL((k_stream<MODE, DEPTH, ROWS>), dim3((1024 / ROWS + 3) / 4, F), dim3(256), 0, 0, d, o, 1024 / ROWS); }); \
	printf("%-44s %.3f ms  %.0f GB/s\n", label, ms, gb / ms * 1e3); }
	RUN(0, 2, 64, "48B-lane rows, depth 2, 64-row strips");
	RUN(0, 3, 64, "48B-lane rows, depth 3, 64-row strips");
	RUN(0, 4, 64, "48B-lane rows, depth 4, 64-row strips");
	RUN(0, 2, 32, "48B-lane rows, depth 2, 32-row strips");
	RUN(0, 4, 16, "48B-lane rows, depth 4, 16-row strips");
	RUN(1, 2, 64, "coalesced 16B lanes, depth 2, 64-row strips");
	RUN(1, 4, 64, "coalesced 16B lanes, depth 4, 64-row strips");
	RUN(1, 4, 16, "coalesced 16B lanes, depth 4, 16-row strips");
	for (int blocks : {2048, 4096, 8192}) {
		float ms = timeit([&] { hipLaunchKernelGGL(k_flat, dim3(blocks), dim3(256), 0, 0, (const uint4*)d, F * FRAME / 16, o); });
		printf("flat grid-stride uint4 x4, %5d blocks          %.3f ms  %.0f GB/s\n", blocks, ms, gb / ms * 1e3);
	}
	return 0;
}
