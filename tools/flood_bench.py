"""GPU: how fast is the exact flood path (K2b)? 64..256 rigidly shifted frames -> every frame is flagged."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcimbar_amd import HipDecoder, framegen, modeb

dev = torch.device("cuda", 0)
synth = framegen.FrameSynth(dev)
dec = HipDecoder(0)
st = torch.cuda.current_stream().cuda_stream
for n in (1, 16, 256, 1024):
    payload = framegen.synth_payload(n, seed=7, device=dev)
    frames = torch.roll(synth.frames_from_payload(payload), shifts=(2, 1), dims=(1, 2)).contiguous()
    chunks = torch.zeros((n, 7500), dtype=torch.uint8, device=dev)
    masks = torch.zeros((n,), dtype=torch.int32, device=dev)
    dec.enable_timing(True)
    for it in range(3):
        dec.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, st)
        torch.cuda.synchronize()
    t = dec.stage_times()
    ok = bool((chunks == payload).all().item()) and bool((masks == 0xFFF).all().item())
    flagged = int(dec.tap(5, n).sum())
    path = dec.tap(7, n)
    print(f"n={n} flagged={flagged} exact={int((path == 1).sum())} batch={int((path == 2).sum())} payload_ok={ok} flood={t['flood']:.2f} ms ({t['flood']/n*1e3:.1f} us/frame, {n/t['flood']*1e3:.0f} frames/s) total={sum(t.values()):.2f} ms")
