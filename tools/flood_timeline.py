"""GPU, under `rocprofv3 --kernel-trace`: nothing special -- BASELINE configs[4] at 1024 captures once (plus a warm-up), so that the trace shows when
the two half-batches' flood kernels start and end. tools/gpu_flood_timeline.sh prints the timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcimbar_amd import HipDecoder
from tools import extractbench
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
st = torch.cuda.current_stream(dev)
n = int(os.environ.get("N", "1024"))
r = extractbench.run(dec, dev, st, None, n=n, reps=1, key="c5")
print({k: r["c5"][k] for k in ("ms", "captures_per_s", "flood_exact_frames")}, flush=True)
