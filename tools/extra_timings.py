"""GPU: secondary timings quoted in DESIGN.md (not the headline bench): host-fed decode (PCIe inclusive), the
should_preprocess path, RS-error-heavy frames (BASELINE configs[2]) and the exact flood path."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcimbar_amd import HipDecoder, framegen, modeb

dev = torch.device("cuda", 0)
synth = framegen.FrameSynth(dev)
dec = HipDecoder(0)
st = torch.cuda.current_stream().cuda_stream


def device_batch(frames, pre=False, reps=5):
    n = frames.shape[0]
    chunks = torch.zeros((n, 7500), dtype=torch.uint8, device=dev); masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    dec.enable_timing(True)
    acc = {}
    for i in range(reps + 2):
        dec.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), pre, 2, st); torch.cuda.synchronize()
        if i >= 2:
            for k, v in dec.stage_times().items(): acc[k] = acc.get(k, 0) + v / reps
    dec.enable_timing(False)
    return acc, chunks, masks


n = 1024
payload = framegen.synth_payload(n, seed=1234, device=dev)
frames = torch.empty((n, 1024, 1024, 3), dtype=torch.uint8, device=dev)
for lo in range(0, n, 64): synth.frames_from_payload(payload[lo:lo + 64], out=frames[lo:lo + 64])

acc, c, m = device_batch(frames)
print("clean      ", {k: round(v, 4) for k, v in acc.items()}, "total %.4f ms" % sum(acc.values()), "ok", bool((c == payload).all()))
acc, c, m = device_batch(frames, pre=True)
print("preprocess ", {k: round(v, 4) for k, v in acc.items()}, "total %.4f ms" % sum(acc.values()), "all chunks", bool((m == 0xFFF).all()))

tiles = framegen.inject_cell_errors(synth.cell_tiles(payload[:256]), n_errors=99, seed=5678)
fe = torch.empty((256, 1024, 1024, 3), dtype=torch.uint8, device=dev)
for lo in range(0, 256, 64): synth.render(tiles[lo:lo + 64], out=fe[lo:lo + 64])
acc, c, m = device_batch(fe)
print("cell errors (256 frames)", {k: round(v, 4) for k, v in acc.items()}, "total %.4f ms" % sum(acc.values()), "ok", bool((c == payload[:256]).all()))

# host-fed: pageable numpy and pinned torch memory, 256 frames, outputs to host as well
h = frames[:256].cpu()
for name, buf in (("pageable", h.numpy()), ("pinned", h.pin_memory().numpy())):
    dec.decode_batch(buf[:8])
    t0 = time.perf_counter(); total, chunks, masks = dec.decode_batch(buf); dt = time.perf_counter() - t0
    print(f"host-fed {name}: 256 frames in {dt*1e3:.1f} ms = {256/dt:.0f} frames/s ({256*3.145728/dt/1e3:.1f} GB/s over PCIe), good bytes {total}")
