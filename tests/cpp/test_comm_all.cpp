// One process driving every GPU of the node through the C library alone -- what a C++ ./cimbar would do: one context + one communicator per
// device (cimbar_hip_comm_init_all), every device decodes its slab of a frame stream, cimbar_hip_gather_chunks (RCCL ncclGather, issued by the
// library) brings the chunk slots to device 0, and the gathered bytes equal what was encoded. Needs >= 2 GPUs (tests/test_gpu_comm_all.py skips
// otherwise); with ndev == 1 it still runs the same calls with a one-rank communicator.
//   test_comm_all ndev frames_per_dev template.bin
#include "../../include/cimbar_hip.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv)
{
	if (argc < 4) { std::printf("usage: test_comm_all ndev frames_per_dev template.bin\n"); return 2; }
	const int ndev = std::atoi(argv[1]), n = std::atoi(argv[2]);
	int have = 0;
	CHECK(hipGetDeviceCount(&have) == hipSuccess && have >= ndev && ndev >= 1 && n >= 1);
	const size_t FR = 1024ull * 1024 * 3, FB = 7500;
	std::vector<unsigned char> tmpl(FR);
	{
		FILE* f = std::fopen(argv[3], "rb");
		CHECK(f && std::fread(tmpl.data(), 1, FR, f) == FR);
		std::fclose(f);
	}
	std::vector<cimbar_hip_ctx*> ctx((size_t)ndev, nullptr);
	std::vector<cimbar_hip_comm*> comm((size_t)ndev, nullptr);
	for (int d = 0; d < ndev; ++d) {
		CHECK(cimbar_hip_create(d, 68, &ctx[d]) == 0);
		CHECK(cimbar_hip_set_template(ctx[d], tmpl.data(), CIMBAR_HIP_MEM_HOST) == 0);
	}
	CHECK(cimbar_hip_comm_init_all(ndev, nullptr, comm.data()) == 0);

	// payload: arbitrary bytes. The decode path interprets fountain headers only for the header-derived colour correction (color_correction 2),
	// which arbitrary bytes would mislead -- so the frames are decoded with color_correction 0 (clean frames need none)
	std::vector<std::vector<unsigned char>> payload((size_t)ndev, std::vector<unsigned char>((size_t)n * FB));
	for (int d = 0; d < ndev; ++d)
		for (size_t k = 0; k < payload[d].size(); ++k) payload[d][k] = (unsigned char)((d * 131 + (k / FB) * 17 + (k % FB) * 7 + (k >> 11)) & 0xFF);

	struct Dev { unsigned char *frames = nullptr, *chunks = nullptr, *all_chunks = nullptr; uint32_t *masks = nullptr, *all_masks = nullptr; hipStream_t st = nullptr; };
	std::vector<Dev> dev((size_t)ndev);
	for (int d = 0; d < ndev; ++d) {
		CHECK(hipSetDevice(d) == hipSuccess);
		CHECK(hipStreamCreate(&dev[d].st) == hipSuccess);
		CHECK(hipMalloc((void**)&dev[d].frames, (size_t)n * FR) == hipSuccess);
		CHECK(hipMalloc((void**)&dev[d].chunks, (size_t)n * FB) == hipSuccess);
		CHECK(hipMalloc((void**)&dev[d].masks, (size_t)n * 4) == hipSuccess);
		if (d == 0) {
			CHECK(hipMalloc((void**)&dev[d].all_chunks, (size_t)ndev * n * FB) == hipSuccess);
			CHECK(hipMalloc((void**)&dev[d].all_masks, (size_t)ndev * n * 4) == hipSuccess);
		}
		CHECK(cimbar_hip_encode_batch(ctx[d], payload[d].data(), n, CIMBAR_HIP_MEM_HOST, dev[d].frames, CIMBAR_HIP_MEM_DEVICE, dev[d].st) == 0);
		CHECK(cimbar_hip_decode_batch(ctx[d], dev[d].frames, n, CIMBAR_HIP_MEM_DEVICE, 0, 0, dev[d].chunks, dev[d].masks, CIMBAR_HIP_MEM_DEVICE, dev[d].st) == 0);
	}
	// "call gather_chunks from one thread per device": RCCL's single-process rule for blocking group semantics
	std::vector<int> rc((size_t)ndev, -100);
	std::vector<std::thread> th;
	for (int d = 0; d < ndev; ++d)
		th.emplace_back([&, d]() {
			(void)hipSetDevice(d);
			rc[d] = cimbar_hip_gather_chunks(ctx[d], comm[d], /*root*/ 0, dev[d].chunks, dev[d].masks, n, dev[d].all_chunks, dev[d].all_masks, dev[d].st);
			if (rc[d] == 0 && hipStreamSynchronize(dev[d].st) != hipSuccess) rc[d] = -101;
		});
	for (auto& t : th) t.join();
	for (int d = 0; d < ndev; ++d) CHECK(rc[d] == 0);

	std::vector<unsigned char> got((size_t)ndev * n * FB);
	std::vector<uint32_t> gm((size_t)ndev * n);
	CHECK(hipSetDevice(0) == hipSuccess);
	CHECK(hipMemcpy(got.data(), dev[0].all_chunks, got.size(), hipMemcpyDeviceToHost) == hipSuccess);
	CHECK(hipMemcpy(gm.data(), dev[0].all_masks, gm.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
	for (int d = 0; d < ndev; ++d) {
		CHECK(std::memcmp(got.data() + (size_t)d * n * FB, payload[d].data(), (size_t)n * FB) == 0);   // rank order == frame order
		for (int f = 0; f < n; ++f) CHECK(gm[(size_t)d * n + f] == 0xFFFu);
	}
	for (int d = 0; d < ndev; ++d) { cimbar_hip_comm_destroy(comm[d]); cimbar_hip_destroy(ctx[d]); }
	std::printf("OK %d device(s) x %d frames gathered on device 0\n", ndev, n);
	return 0;
}
