"""PNG ingest rate against the size of the host pool (tools: run on the GPU box)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from libcimbar_amd import HipDecoder, framegen, ingest
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
payload = framegen.synth_payload(128, seed=5151, device=dev)
fr = torch.empty((128, 1024, 1024, 3), dtype=torch.uint8, device=dev)
dec.encode_batch_device(payload.data_ptr(), 128, fr.data_ptr())
torch.cuda.synchronize()
host = fr.cpu().numpy()
print("host threads", os.cpu_count())
with tempfile.TemporaryDirectory() as td:
    paths = []
    for k in range(128):
        p = os.path.join(td, f"f{k:03d}.png"); Image.fromarray(host[k]).save(p, compress_level=1); paths.append(p)
    paths = paths * 8
    for th in (8, 16, 32, 64, 0):          # 0 = the library default (the CPUs the process may use)
        ing = ingest.Ingest(dec, threads=th, batch_frames=64, ring=3)
        ing.run_files(paths[:128])
        t0 = time.perf_counter(); total, c, m = ing.run_files(paths); dt = time.perf_counter() - t0
        print("threads", th, "frames/s", round(len(paths) / dt, 1), "ok", total == len(paths) * 7500, ing.timings())
        ing.close()
