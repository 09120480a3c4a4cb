"""GPU: BASELINE configs[4] rows only (256 and 1024 captures), without the rest of bench.py's extras."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcimbar_amd import HipDecoder
from tools import extractbench
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
st = torch.cuda.current_stream(dev)
for n, reps in ((256, 3), (1024, 1)):
    r = extractbench.run(dec, dev, st, None, n=n, reps=reps, key=f"config5_{n}")
    v = list(r.values())[0]
    print(n, {k: v[k] for k in ("ms", "captures_per_s", "extract_only_ms", "frames_fully_decoded", "payload_ok_where_decoded", "flood_exact_frames")}, flush=True)
    torch.cuda.empty_cache()
