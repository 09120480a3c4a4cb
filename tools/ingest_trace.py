"""Workload for tools/gpu_profile_ingest.sh: 512 page-locked host frames through cimbar_ingest_run_raw (H2D on the copy stream, pipelined decode,
D2H of the chunks), three times."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from libcimbar_amd import HipDecoder, framegen, ingest  # noqa: E402

dev = torch.device("cuda", 0)
dec = HipDecoder(0)
n = 128
payload = framegen.synth_payload(n, seed=5151, device=dev)
fr = torch.empty((n, 1024, 1024, 3), dtype=torch.uint8, device=dev)
dec.encode_batch_device(payload.data_ptr(), n, fr.data_ptr())
torch.cuda.synchronize()
host = torch.from_numpy(np.ascontiguousarray(np.tile(fr.cpu().numpy(), (4, 1, 1, 1)))).pin_memory()
ing = ingest.Ingest(dec, threads=0, batch_frames=64, ring=3)
ing.run_raw_ptr(host.data_ptr(), 64)
for _ in range(3):
    t0 = time.perf_counter()
    total = ing.run_raw_ptr(host.data_ptr(), 512)
    dt = time.perf_counter() - t0
    print("frames/s", 512 / dt, "good", total == 512 * 7500)
ing.close()
