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
