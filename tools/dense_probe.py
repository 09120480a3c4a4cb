"""GPU: the exact flood replay at four vs eight frames per CU (CIMBAR_HIP_FLOOD_DENSE) on BASELINE configs[4] captures, single batches and the
two-context stream, optionally over several builds of the library (different LDS heap sizes of the dense instance).
Usage: python tools/dense_probe.py [lib=path ...]   (worker: python tools/dense_probe.py --worker n contexts)"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch
    from libcimbar_amd import HipDecoder
    from tools import extractbench
    n, contexts = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    if contexts <= 1:
        dec = HipDecoder(0)
        r = extractbench.run(dec, dev, torch.cuda.current_stream(dev), None, n=n, reps=2, key="r", fmt=int(os.environ.get("PROBE_FMT", "3")))["r"]
        print("ROW", json.dumps({k: r[k] for k in ("captures", "ms", "captures_per_s", "extract_only_ms", "frames_fully_decoded", "payload_ok_where_decoded", "flood_exact_frames")}))
    else:
        r = extractbench.run_stream(dev, n=n, contexts=contexts, batches=6, reps=2, key="r")["r"]
        print("ROW", json.dumps({k: r[k] for k in ("captures_per_batch", "contexts", "ms_per_batch", "captures_per_s", "frames_fully_decoded_per_batch", "payload_ok_where_decoded")}))
    sys.exit(0)

# plan entries: dense:grid:n:contexts
spec = [a for a in sys.argv[1:] if a.count(":") == 3] or ["0:2048:1024:1", "0:2048:2048:1", "1:2048:1024:1", "1:2048:2048:1", "1:2048:2048:2"]
out = {}
for item in spec:
    dense, grid, n, contexts = (int(x) for x in item.split(":"))
    env = dict(os.environ, CIMBAR_HIP_FLOOD_DENSE=str(dense), CIMBAR_HIP_FLOOD_DENSE_GRID=str(grid))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(n), str(contexts)], env=env, capture_output=True, text=True, timeout=600)
    rows = [l[4:] for l in p.stdout.splitlines() if l.startswith("ROW ")]
    key = f"dense={dense}|grid={grid}|{n}x{contexts}"
    out[key] = json.loads(rows[-1]) if rows else {"error": (p.stderr or p.stdout)[-400:]}
    print(key, json.dumps(out[key]), flush=True)
json.dump(out, open(os.environ.get("PROBE_OUT", "gpurun_out/dense_probe.json"), "w"), indent=1)
