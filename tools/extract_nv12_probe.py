"""GPU: bench.py's configs[4] row on NV12 captures (format 12), n per batch from argv (default 2048): the chain's wall time + the extract stage alone.
With CIMBAR_HIP_WARP_TWOPASS=0 the warp converts each tap itself (k_warp<12>) instead of k_roi_boxes + k_convert_roi + k_warp<3>."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libcimbar_amd import HipDecoder
from tools import extractbench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
r = extractbench.run(dec, dev, torch.cuda.current_stream(dev), None, n=n, reps=2, key="nv12", fmt=12)["nv12"]
print(json.dumps({k: r[k] for k in ("captures", "format", "ms", "captures_per_s", "extract_only_ms", "frames_fully_decoded", "payload_ok_where_decoded")} | {"twopass": os.environ.get("CIMBAR_HIP_WARP_TWOPASS", "1")}))
