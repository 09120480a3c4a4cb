"""GPU: per-kernel stage times of one 1024-frame clean batch for color_correction 2 / 0 and the sharpen variant (dev aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcimbar_amd import HipDecoder, framegen, modeb
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
st = torch.cuda.current_stream().cuda_stream
n = 1024
payload = framegen.synth_payload(n, seed=1, device=dev)
frames = torch.empty((n, 1024, 1024, 3), dtype=torch.uint8, device=dev)
dec.encode_batch_device(payload.data_ptr(), n, frames.data_ptr(), st)
chunks = torch.zeros((n, 7500), dtype=torch.uint8, device=dev); masks = torch.zeros((n,), dtype=torch.int32, device=dev)
dec.enable_timing(True)
for cc in (2, 0):
    acc = {}
    for i in range(6):
        dec.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, cc, st); torch.cuda.synchronize()
        if i:
            for k, v in dec.stage_times().items(): acc[k] = acc.get(k, 0) + v / 5
    print("cc", cc, {k: round(v, 4) for k, v in acc.items()}, "sum", round(sum(acc.values()), 4), "ok", bool((chunks == payload).all()))
