"""GPU: headline step and BASELINE configs[2] (99 substituted cells per frame) with stage times -- the quick A/B for chain-kernel changes
(bench.py's own helpers, nothing else of its run)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from libcimbar_amd import HipDecoder, framegen, modeb

dev = torch.device("cuda", 0)
dec = HipDecoder(0)
synth = framegen.FrameSynth(dev)
stream = torch.cuda.current_stream(dev)
n, steps = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 100
outs = [(torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros((n,), dtype=torch.int32, device=dev)) for _ in range(4)]
for name, errs in (("clean", 0), ("cell_errors", 99)):
    inputs, payloads = [], []
    for k in range(4 if not errs else 1):
        payload = framegen.synth_payload(n, seed=1234 + k, device=dev)
        tiles = synth.cell_tiles(payload)
        if errs:
            tiles = framegen.inject_cell_errors(tiles, n_errors=errs, seed=5678)
        fe = torch.empty((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
        for lo in range(0, n, 64):
            synth.render(tiles[lo:lo + 64], out=fe[lo:lo + 64])
        inputs.append(fe); payloads.append(payload)
        del tiles
    ms = sorted(bench.stream_ms(dec, inputs, outs, steps, 8, True, stream, dev) for _ in range(3))[1]
    ms_u = bench.stream_ms(dec, inputs, outs, max(8, steps // 4), 2, False, stream, dev)
    st = bench.stage_times(dec, inputs[0], outs[0], stream, dev, reps=4)
    ok = bool((outs[0][1] == 0xFFF).all().item()) and bool((outs[0][0] == payloads[0]).all().item())
    print(name, json.dumps({"ms_per_step": round(ms, 4), "no_pipeline_ms_per_step": round(ms_u, 4), "payload_ok": ok, "stage_ms": {k: round(v, 4) for k, v in st.items()}}), flush=True)
    del inputs
    torch.cuda.empty_cache()
