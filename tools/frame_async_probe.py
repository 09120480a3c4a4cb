"""Where a synchronous / overlapped single-frame call spends its time (host clock around the two C calls)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from libcimbar_amd import HipDecoder, framegen
dec = HipDecoder(0)
synth = framegen.FrameSynth("cpu")
payload = framegen.synth_payload(16, seed=5)
pinned_t = synth.frames_from_payload(payload).pin_memory()
pinned = pinned_t.numpy()
pageable = np.array(pinned)
for name, fr in (("pinned", pinned), ("pageable", pageable), ("pinned", pinned), ("pageable", pageable)):
    for q in range(6): dec.decode_frame(fr[q])
    ta = tw = 0.0
    t0 = time.perf_counter()
    for q in range(64):
        a = time.perf_counter()
        t = dec.decode_frame_async(fr[q % 16])
        b = time.perf_counter()
        dec.decode_frame_wait(t)
        c = time.perf_counter()
        ta += b - a; tw += c - b
    print(f"sync {name}: {(time.perf_counter() - t0) / 64 * 1e3:.4f} ms per frame (async call {ta / 64 * 1e3:.4f}, wait {tw / 64 * 1e3:.4f})", flush=True)
depth = dec.pipeline_depth
for name, fr in (("pinned", pinned), ("pageable", pageable)):
    for rep in range(2):
        tk = []
        t0 = time.perf_counter()
        for q in range(256):
            tk.append(dec.decode_frame_async(fr[q % 16]))
            if len(tk) >= depth: dec.decode_frame_wait(tk.pop(0))
        while tk: dec.decode_frame_wait(tk.pop(0))
        dt = (time.perf_counter() - t0) / 256
        print(f"in flight {name}: {dt * 1e3:.4f} ms per frame = {1 / dt:.0f} frames/s", flush=True)
