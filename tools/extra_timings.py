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

# the stage in front of the decoder: 1080p captures resident in HBM -> binary image for the scanner; -> deskewed 1024x1024 frames
import ctypes
from libcimbar_amd import decoder as D
ncap, W, H = 256, 1920, 1080
caps = torch.zeros((ncap, H, W, 3), dtype=torch.uint8, device=dev)
caps[:, 28:1052, 448:1472] = frames[:ncap]                    # the frame pasted upright into a dark capture
binimg = torch.empty((ncap, H, W), dtype=torch.uint8, device=dev)
out = torch.empty((ncap, 1024, 1024, 3), dtype=torch.uint8, device=dev)
corners = np.tile(np.array([478, 58, 1442, 58, 478, 1022, 1442, 1022], np.float32), (ncap, 1))   # anchor centres of that paste
lib = D.load_library()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for it in range(3):
    ev[0].record()
    rc = lib.cimbar_hip_scan_preprocess(dec._ctx, ctypes.c_void_p(caps.data_ptr()), W, H, ncap, 1, ctypes.c_void_p(binimg.data_ptr()), None, 1, None)
    ev[1].record()
    dec.deskew_batch_device(caps.data_ptr(), W, H, ncap, corners, out.data_ptr(), None)
    ev[2].record()
    torch.cuda.synchronize()
t_pre, t_warp = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
chunks = torch.zeros((ncap, 7500), dtype=torch.uint8, device=dev); masks = torch.zeros((ncap,), dtype=torch.int32, device=dev)
dec.decode_batch_device(out.data_ptr(), ncap, chunks.data_ptr(), masks.data_ptr(), False, 2, None); torch.cuda.synchronize()
print(f"extract stage, {ncap} captures 1920x1080: scan_preprocess {t_pre:.3f} ms ({ncap/t_pre*1e3:.0f} captures/s, "
      f"{ncap*(W*H*4)/t_pre/1e6:.0f} GB/s rd+wr), deskew {t_warp:.3f} ms ({ncap/t_warp*1e3:.0f} captures/s); "
      f"deskewed frames decode: {int((masks == 0xFFF).sum())}/{ncap} complete, payload ok {bool((chunks == payload[:ncap]).all())}")
