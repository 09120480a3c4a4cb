"""GPU: where a step of the exact flood replay (k_flood3) spends its cycles, per wavefront -- needs the -DFLOOD_PROF build of the library
(CIMBAR_HIP_LIB=tools/_prof/libcimbar_hip_prof.so). Config-5 captures (deskewed camera frames) and rigidly shifted frames, at 256 and 1024
frames per batch. Prints cycles per step for the heap owner H (hand-over, pushes, pop, waiting) and the decoder D (hand-over, window fetch,
decode + offers, waiting)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libcimbar_amd import HipDecoder, framegen, modeb
from tools import extractbench

dev = torch.device("cuda", 0)
dec = HipDecoder(0)
st = torch.cuda.current_stream(dev)
out = {}
KINDS = tuple(os.environ.get("PROF_KINDS", "config5,shift").split(","))
SIZES = tuple(int(x) for x in os.environ.get("PROF_NS", "256,1024").split(","))      # (2048 and up: the dense instance, eight frames per CU)
for kind in KINDS:
    for n in SIZES:
        payload = framegen.synth_payload(n, seed=777, device=dev)
        frames = torch.empty((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
        dec.encode_batch_device(payload.data_ptr(), n, frames.data_ptr(), st.cuda_stream)
        chunks = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
        masks = torch.zeros((n,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        best = None
        for rep in range(2):
            if kind == "config5":
                caps = extractbench.make_captures(frames)
                status = torch.zeros((n,), dtype=torch.int32, device=dev)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                dec.scan_extract_decode_device(caps.data_ptr(), 1920, 1080, n, chunks.data_ptr(), masks.data_ptr(), status.data_ptr(), -1, 2, st.cuda_stream)
            else:
                sh = torch.roll(frames, shifts=(2, 1), dims=(1, 2)).contiguous()
                os.environ["X"] = "1"
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                dec.decode_batch_device(sh.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, st.cuda_stream)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        raw = np.zeros((n, 16), np.uint64)
        rc = dec._lib.cimbar_hip_tap(dec._ctx, 100, raw.ctypes.data_as(ctypes.c_void_p), raw.nbytes)
        if rc < 0:
            print("tap 100 not available: not the FLOOD_PROF build"); sys.exit(1)
        path = dec.tap(7, n)
        hw_h, hw_d = (raw[:, 6] >> np.uint64(32)).astype(np.int64), (raw[:, 14] >> np.uint64(32)).astype(np.int64)   # HW_ID of either wavefront
        raw[:, 6] &= np.uint64(0xFFFFFFFF)
        raw[:, 14] &= np.uint64(0xFFFFFFFF)
        keep = raw[:, 6] > 0
        use = raw[keep]
        steps = use[:, 6].astype(np.float64)
        simd = lambda h: (h >> 4) & 3
        slot = lambda h: h & 15
        cu = lambda h: ((h >> 8) & 15) | (((h >> 12) & 1) << 4) | (((h >> 13) & 7) << 5)      # CU_ID | SH_ID | SE_ID (per XCC)
        import collections
        placement = collections.Counter((int(simd(a)), int(simd(b)), int(slot(a)), int(slot(b))) for a, b in zip(hw_h[keep], hw_d[keep]))
        per_simd = collections.Counter()
        for a in hw_h[keep]:
            per_simd[(int(cu(a)), int(simd(a)))] += 1
        h_share = collections.Counter(per_simd.values())          # how many heap owners share one SIMD of one (XCC-local) CU id
        H = {k: float((use[:, i] / steps).mean()) for k, i in (("handover", 0), ("pushes", 1), ("pop", 2), ("wait", 5))}
        Dv = {k: float((use[:, 8 + i] / use[:, 14].astype(np.float64)).mean()) for k, i in (("handover", 0), ("fetch", 3), ("decode_offers", 4), ("wait", 5))}
        row = {"frames": n, "ms": round(best * 1e3, 2), "exact_frames": int((path == 1).sum()), "areas_with_data": int(len(use)), "steps_per_frame": float(steps.mean()),
               "H_cycles_per_step": {k: round(v, 1) for k, v in H.items()}, "D_cycles_per_step": {k: round(v, 1) for k, v in Dv.items()},
               "H_total": round(sum(H.values()), 1), "D_total": round(sum(Dv.values()), 1),
               "own_stale_pops_per_frame": float(use[:, 7].mean()),
               "placement_simdH_simdD_slotH_slotD": {str(k): v for k, v in placement.most_common(12)},
               "heap_owners_per_(cu_id,simd)_histogram": {str(k): v for k, v in sorted(h_share.items())}}
        out[f"{kind}_{n}"] = row
        print(kind, n, json.dumps(row), flush=True)
        torch.cuda.empty_cache()
json.dump(out, open(os.environ.get("FLOOD_PROF_OUT", "gpurun_out/flood3_prof.json"), "w"), indent=1)
