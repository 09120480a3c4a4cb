// `./cimbar img1.png img2.png ... -o out_dir --no-deskew` (src/exe/cimbar/cimbar.cpp:124-171,279-296) with the MI355X path underneath:
// libcimbar_ingest's PNG pool + pinned ring feeds cimbar_hip_decode_batch_pipelined, the chunks go to the REFERENCE's fountain_decoder_sink
// with its decompress_on_store writer -- the file appears in out_dir exactly as the reference CLI would write it.
// Built in the build container against the reference headers (oracle/Makefile `dropin`), run on the GPU box by tests/test_gpu_ingest.py.
//
//   cimbar_amd_cli [--device-png] out_dir img1.png [img2.png ...]
// --device-png: the PNGs are inflated and un-filtered on the GPU too (cimbar_ingest_create_ex, CIMBAR_INGEST_PNG_DEVICE)
#include "cimb_translator/Config.h"
#include "compression/zstd_decompressor.h"
#include "fountain/fountain_decoder_sink.h"

#include "../../include/cimbar_ingest.h"

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

namespace {
struct SinkState { fountain_decoder_sink* sink; unsigned long long chunks = 0; };

int feed(void* user, const uint8_t* chunks, const uint32_t* masks, int first_frame, int n)
{
	(void)first_frame;
	SinkState* st = (SinkState*)user;
	for (int f = 0; f < n; ++f)
		for (int j = 0; j < CIMBAR_HIP_CHUNKS_PER_FRAME; ++j)
			if (masks[f] & (1u << j)) {
				st->sink->write((const char*)chunks + ((size_t)f * CIMBAR_HIP_CHUNKS_PER_FRAME + j) * CIMBAR_HIP_CHUNK_SIZE, CIMBAR_HIP_CHUNK_SIZE);
				st->chunks++;
			}
	return 0;
}
}

int main(int argc, char** argv)
{
	const bool device_png = argc > 1 && std::string(argv[1]) == "--device-png";
	if (device_png) { ++argv; --argc; }
	if (argc < 3) { std::printf("usage: cimbar_amd_cli [--device-png] out_dir img1.png [img2.png ...]\n"); return 2; }
	cimbar::Config::update(68);
	cimbar_hip_ctx* ctx = nullptr;
	if (cimbar_hip_create(0, 68, &ctx) != 0) { std::printf("no device\n"); return 3; }
	cimbar_ingest* ing = nullptr;
	if (cimbar_ingest_create_ex(ctx, 0, device_png ? 256 : 16, 3, device_png ? CIMBAR_INGEST_PNG_DEVICE : CIMBAR_INGEST_PNG_HOST, 0, &ing) != 0) { std::printf("ingest create failed\n"); return 3; }
	fountain_decoder_sink sink(cimbar::Config::fountain_chunk_size(), decompress_on_store<std::ofstream>(argv[1], true));
	SinkState st{&sink};
	const long long good = cimbar_ingest_run_files(ing, argv + 2, argc - 2, 0, 2, feed, &st);
	std::printf("good bytes %lld, chunks %llu, files done %u\n", good, st.chunks, sink.num_done());
	cimbar_ingest_destroy(ing);
	cimbar_hip_destroy(ctx);
	return sink.num_done() >= 1 ? 0 : 4;      // (the reference CLI's exit code is an error mask; 0 = everything decoded)
}
