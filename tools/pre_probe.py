import sys, os
sys.path.insert(0, "/root/repo")
import torch
from libcimbar_amd import HipDecoder, framegen, modeb
import bench
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
payload, fr = bench.make_frames(1024, dev, 99, dec, check=False)
outs = (torch.zeros((1024, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros((1024,), dtype=torch.int32, device=dev))
st = torch.cuda.current_stream(dev)
t = bench.stage_times(dec, fr, outs, st, dev, reps=5, pre=True)
print(os.environ.get("CIMBAR_HIP_K1_STRIPS", "auto"), "k_threshold<3,true> per 1024 frames:", round(t["threshold"], 4), "ms")
