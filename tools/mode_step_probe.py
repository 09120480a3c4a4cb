"""GPU: pipelined ms per 1024-frame step of one mode (and the threshold kernel alone), for A/B runs over library builds. Usage: python tools/mode_step_probe.py <mode>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from libcimbar_amd import HipDecoder, framegen, geometry
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 67
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
g = geometry.for_mode(mode)
d = HipDecoder(0, mode)
n = 1024
payload = framegen.synth_payload(n, seed=6767, device=dev, mode=mode)
f = torch.empty((n, *g.FRAME_SHAPE), dtype=torch.uint8, device=dev)
d.encode_batch_device(payload.data_ptr(), n, f.data_ptr())
o = [(torch.zeros((n, g.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(d.pipeline_depth)]
ms = [bench.stream_ms(d, [f], o, 100, 4, True, st, dev) for _ in range(3)]
ok = all(bool((m == g.FULL_MASK).all().item()) and bool((c == payload).all().item()) for c, m in o)
t = bench.stage_times(d, f, o[0], st, dev, reps=3)
print("mode", mode, os.environ.get("CIMBAR_HIP_K1_STRIPS", "auto"), os.path.basename(os.environ.get("CIMBAR_HIP_LIB", "default")), "pipelined ms per step", [round(x, 4) for x in ms], "K1 alone", round(t["threshold"], 4), "ok", ok)
