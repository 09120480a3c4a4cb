import sys, time, json
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from libcimbar_amd import HipDecoder, framegen
dec = HipDecoder(0)
synth = framegen.FrameSynth("cpu")
payload = framegen.synth_payload(16, seed=5)
frames = synth.frames_from_payload(payload).pin_memory().numpy()
for q in range(8): dec.decode_frame(frames[q])
t0 = time.perf_counter()
for q in range(64): r, c, m = dec.decode_frame(frames[q % 16])
dt = (time.perf_counter() - t0) / 64
print("single frame (pinned host in):", round(dt * 1e3, 4), "ms", r, hex(m))
fr2 = np.array(frames)   # pageable
for q in range(4): dec.decode_frame(fr2[q])
t0 = time.perf_counter()
for q in range(64): r, c, m = dec.decode_frame(fr2[q % 16])
print("single frame (pageable host in):", round((time.perf_counter() - t0) / 64 * 1e3, 4), "ms")
dec.enable_timing(True)
dev = torch.device("cuda", 0)
d = torch.from_numpy(frames[:1]).to(dev)
ch = torch.zeros((1, 7500), dtype=torch.uint8, device=dev); ms = torch.zeros((1,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for q in range(4):
    dec.decode_batch_device(d.data_ptr(), 1, ch.data_ptr(), ms.data_ptr(), False, 2, st); torch.cuda.synchronize()
print("stage times n=1:", {k: round(v, 4) for k, v in dec.stage_times().items()})
dec.enable_timing(False)
t0 = time.perf_counter()
for q in range(64):
    dec.decode_batch_device(d.data_ptr(), 1, ch.data_ptr(), ms.data_ptr(), False, 2, st); torch.cuda.synchronize()
print("device-resident n=1 call+sync:", round((time.perf_counter() - t0) / 64 * 1e3, 4), "ms")
# ---- what the call is made of: the H2D copy alone, the two D2H copies alone, a stream sync behind nothing
hf = torch.from_numpy(frames[:1])          # pinned
dd = torch.empty_like(d)
for q in range(4): dd.copy_(hf, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for q in range(64): dd.copy_(hf, non_blocking=True); torch.cuda.synchronize()
print("H2D 3 MB pinned copy+sync:", round((time.perf_counter() - t0) / 64 * 1e3, 4), "ms")
hc = torch.empty((1, 7500), dtype=torch.uint8).pin_memory(); hm = torch.empty((1,), dtype=torch.int32).pin_memory()
for q in range(4): hc.copy_(ch, non_blocking=True); hm.copy_(ms, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for q in range(64): hc.copy_(ch, non_blocking=True); hm.copy_(ms, non_blocking=True); torch.cuda.synchronize()
print("two D2H copies + sync:", round((time.perf_counter() - t0) / 64 * 1e3, 4), "ms")
t0 = time.perf_counter()
for q in range(64): hm.copy_(ms, non_blocking=True); torch.cuda.synchronize()
print("one 4-byte D2H + sync:", round((time.perf_counter() - t0) / 64 * 1e3, 4), "ms")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for q in range(64): dec.decode_batch_device(d.data_ptr(), 1, ch.data_ptr(), ms.data_ptr(), False, 2, st)
e1.record(); torch.cuda.synchronize()
print("device-resident n=1, 64 calls back to back, GPU time per call:", round(e0.elapsed_time(e1) / 64, 4), "ms")
# ---- frames in flight
dec.enable_timing(False)
depth = dec.pipeline_depth
for rep in range(3):
    tk = []
    t0 = time.perf_counter()
    for q in range(256):
        tk.append(dec.decode_frame_async(frames[q % 16]))
        if len(tk) >= depth: dec.decode_frame_wait(tk.pop(0))
    while tk: dec.decode_frame_wait(tk.pop(0))
    dt = (time.perf_counter() - t0) / 256
    print("frames in flight (pinned):", round(dt * 1e3, 4), "ms per frame =", round(1 / dt), "frames/s")
t0 = time.perf_counter()
for q in range(64): tk = dec.decode_frame_async(frames[q % 16])
print("host time of one async call (no wait, ring full -> includes completing the oldest):", round((time.perf_counter() - t0) / 64 * 1e3, 4), "ms")
for q in range(4): dec.decode_frame_wait(tk - q) if q < depth else None
