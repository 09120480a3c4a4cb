#!/usr/bin/env python3
"""bench.py -- decoded cimbar frames/s (1024x1024 mode B) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--config 2|4]

One "step" = one pass of the whole decode path (threshold -> symbols -> RS -> CCM -> colours -> RS -> chunks) over one
device-resident batch of F synthetic clean mode-B frames (BASELINE.json configs[1]: F = 1024). Consecutive steps decode
DIFFERENT batches (D = pipeline depth distinct inputs are rotated), so that steps in flight never share input cache lines.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank decodes its own F frames (weak scaling) and the
step ends with the gather of the decoded chunks to rank 0 over RCCL. Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline      dominant kernel's ALGORITHMIC bytes/s (3 153 232 B per frame) vs 8 TB/s HBM, timed live
                with HIP events on the launch stream inside the library
  cpu_baseline  the CPU decoder timed on this host on a bounded sample of the same frames ("reference" = the
                reference's own sources built into oracle/_ref, else "port" = oracle/cimbar_oracle.c); its chunks' SHA-256
                is compared with the GPU's chunks for the same frames (payload_sha_match)
  no_pipeline   the same steps through the ordinary (un-pipelined) entry point
  extra         N = 1 only: BASELINE configs[2] (99 substituted cells per frame), host-fed decode (pinned host frames in,
                H2D inside the timed region), the extractor chain of configs[4] (1080p captures -> scan -> deskew -> decode)
--config 4 runs BASELINE configs[3] instead (strong scaling: an 8192-frame fountain stream of a 16 MiB file split over the
ranks, gather, rank-0 wirehair sink fed inside the timed region): see bench_config4.py.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from libcimbar_amd import HipDecoder, framegen, modeb, multigpu  # noqa: E402

ALGO_BYTES_PER_FRAME = modeb.FRAME_RGB_BYTES + modeb.FRAME_BYTES + 4   # 3 153 232, SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0                                                   # MI355X_MICROARCH.md: 8 TB/s spec


def make_frames(n, device, seed, dec, check=True):
    """Synthetic clean mode-B frames, rendered on the device by the library's encode half (cimbar_hip_encode_batch, checked
    byte-for-byte against the reference encoder in tests/test_gpu_encode.py); 16 of them are cross-checked here against the
    torch restatement of Encoder::encode_next."""
    payload = framegen.synth_payload(n, seed=seed, device=device)
    frames = torch.empty((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=device)
    dec.encode_batch_device(payload.data_ptr(), n, frames.data_ptr(), torch.cuda.current_stream(device).cuda_stream)
    if check:
        k = min(n, 16)
        if not bool((frames[:k] == framegen.FrameSynth(device).frames_from_payload(payload[:k])).all().item()):
            raise SystemExit("bench: device-rendered frames differ from the reference layout")
    return payload, frames


def measured_traffic(kernel, n, tall=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC run of this same command (tools/gpu_profile.sh ->
    profiles/*_pmc_summary.json): FETCH_SIZE (KiB, x2: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md) +
    WRITE_SIZE (KiB), scaled to n frames. None if no PMC summary has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    # (round 5: the threshold kernel has two instances, short strips <2, false, 2> and tall ones <2, false, 7>; older summaries name it <2, false>)
    # (the strip heights are build parameters -- -DCIMBAR_K1_TALLROWS_B, K1_CELLROWS --: the tall instance is the `k_threshold<2, false, N>` with the largest N
    # a summary holds, the short one the smallest; the names below are the default build's and come first)
    keys = {"threshold": ["k_threshold<2, false, 7>"] if tall else ["k_threshold<2, false, 2>", "k_threshold<2, false>"]}.get(kernel)
    import re

    def by_rows(pmc):
        rows = sorted((int(m.group(1)), k) for k in pmc for m in [re.fullmatch(r"k_threshold<2, false, (\d+)>", k)] if m)
        return [rows[-1][1] if tall else rows[0][1]] if rows else []
    key = {"threshold": "k_threshold<2, false>", "symbols": "k_symbols", "rs_symbols": "k_rs<4>", "rs_colors": "k_rs<2>",
           "colors": "k_colors", "frame_mid": "k_frame_mid", "frame_end": "k_frame_end", "flood": "k_flood3"}[kernel]
    for fallback in (False, True):     # (first the default build's kernel names in any summary, only then "largest / smallest N" of a non-default build's)
        for path in reversed(files):      # the newest summary that holds this kernel's counters (summaries of other commands live there too)
            try:
                pmc = json.load(open(path))["pmc"]
                names = by_rows(pmc) if (fallback and kernel == "threshold") else (keys or [key])
                c = next(pmc[k] for k in names if k in pmc)
                per_1024 = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                return per_1024 * n / 1024.0, os.path.basename(path)
            except Exception:
                continue
    return None, None


def trace_average(kernel, tall=False):
    """Average duration (us) of `kernel` in the newest committed rocprofv3 kernel trace of this command that holds it (profiles/*_kernel_stats.csv,
    written by tools/gpu_profile.sh from `rocprofv3 --kernel-trace --stats`): the figure the HIP-event timing in the same line must agree with."""
    import csv
    import glob
    if kernel != "threshold":
        return None, None
    want = "k_threshold<2, false, 7>" if tall else "k_threshold<2, false, 2>"
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")) if "pipelined" not in os.path.basename(f)]
    for path in sorted(files, key=lambda f: os.path.basename(f), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                if row["kernel"] == want:
                    return round(float(row["avg_ns"]) / 1e3, 2), os.path.basename(path)
        except Exception:
            continue
    return None, None


def usable_cpus():
    """CPUs this process may really use: the scheduler affinity, cut down to the container's CPU quota where there is one (cgroup v2 cpu.max).
    The GPU boxes of the pool report 256 hardware threads and run under a 16-CPU quota: threads beyond the quota only get throttled."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(frames_host, gpu_chunks, budget_s=15.0):
    """Time the CPU decoder on host cores over a bounded sample of the same frames, and compare what it decodes with the
    GPU's chunks for those frames (SHA-256 over the sample's 12 x 625-byte chunk slots, frame order)."""
    from oracle import pyref
    import ctypes
    ref = pyref.ref_lib()
    kind = "reference" if ref is not None else "port"
    orc = pyref.oracle_lib()
    hw = os.cpu_count() or 1
    ncores = usable_cpus()
    threads = max(1, min(ncores, 64, len(frames_host)))

    def decode_one(fr, state, chunks=None):
        if chunks is None:
            chunks = np.zeros(7500, np.uint8)
        mask = ctypes.c_uint32(0)
        if ref is not None:
            return ref.ref_decode_fountain(pyref.P(fr), 1024, 1024, 0, 2, 0, pyref.P(chunks), ctypes.byref(mask))
        return orc.co_decode_fountain(pyref.P(fr), 1024, 1024, 0, 2, ctypes.byref(state), pyref.P(chunks), ctypes.byref(mask))

    # what the CPU decoder makes of the sample, frame by frame on one thread (CCM carried in frame order, like the GPU batch)
    st = pyref.CoCcm()
    if ref is not None:
        ref.ref_reset_ccm()
    cpu_chunks = np.zeros((len(frames_host), 7500), np.uint8)
    t0 = time.perf_counter()
    for k in range(len(frames_host)):
        if decode_one(frames_host[k], st, cpu_chunks[k]) != 7500:
            raise RuntimeError("cpu baseline decoded a clean frame incorrectly")
    per_frame = (time.perf_counter() - t0) / len(frames_host)
    sha_cpu = hashlib.sha256(cpu_chunks.tobytes()).hexdigest()
    sha_gpu = hashlib.sha256(np.ascontiguousarray(gpu_chunks).tobytes()).hexdigest()

    total = int(max(threads, 0.6 * budget_s / per_frame * threads))   # threads contend for memory bandwidth: ~60 % of ideal scaling
    done = [0] * threads

    def worker(tid):
        state = pyref.CoCcm()
        for k in range(tid, total, threads):
            if decode_one(frames_host[k % len(frames_host)], state) != 7500:
                raise RuntimeError("cpu baseline decoded a clean frame incorrectly")
            done[tid] += 1

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(sum(done) / dt, 2), "unit": "frames/s", "cores": ncores, "threads": threads, "host_hw_threads": hw, "kind": kind,
            # what stands in for OpenCV under the reference's sources: oracle/cvshim, a scalar restatement of the dozen cv:: calls they make --
            # the reference's control flow and data structures at -O2, but NOT OpenCV's SIMD kernels (no OpenCV exists in this image)
            "opencv": "cv-shim (scalar)" if kind == "reference" else "none (C port)",
            "sample": f"{sum(done)} decodes of {len(frames_host)} bench frames, {threads} threads x 1 Decoder, {dt:.1f} s wall",
            "single_thread_frames_per_s": round(1.0 / per_frame, 1), "wall_s": round(dt, 2),
            "payload_sha_match": sha_cpu == sha_gpu, "payload_sha256": sha_cpu, "payload_frames_compared": len(frames_host)}


def cpu_baseline_config5(sample, budget_s=10.0):
    """BASELINE configs[4] on the host cores: the reference's Extractor::extract + Decoder::decode_fountain (oracle/_ref; `preprocess` as
    cimbar.cpp:147-154 guesses it from the extractor's verdict) over a bounded sample of the SAME 1080p captures, one decoder per thread, and
    its chunks compared with what the GPU delivered for those captures."""
    from oracle import pyref
    import ctypes
    ref = pyref.ref_lib()
    if ref is None:
        return {"error": "oracle/_ref not built: no reference Extractor to time"}
    caps, g_chunks, g_masks = sample["captures"], sample["chunks"], sample["masks"]
    k, h, w = caps.shape[0], caps.shape[1], caps.shape[2]
    ncores = usable_cpus()
    threads = max(1, min(ncores, 64, k))

    def one(cap, chunks):
        frame = np.zeros((1024, 1024, 3), np.uint8)
        mask = ctypes.c_uint32(0)
        res = ref.ref_extract(pyref.P(cap), w, h, pyref.P(frame))
        if not res:
            chunks[:] = 0
            return 0
        ref.ref_decode_fountain(pyref.P(frame), 1024, 1024, 1 if res == 2 else 0, 2, 0, pyref.P(chunks), ctypes.byref(mask))
        return mask.value

    ref.ref_reset_ccm()
    cpu_chunks = np.zeros((k, 7500), np.uint8)
    cpu_masks = np.zeros(k, np.uint32)
    t0 = time.perf_counter()
    for i in range(k):                       # frame order on one thread: the CCM carries over like in the GPU batch
        cpu_masks[i] = one(caps[i], cpu_chunks[i])
    per = (time.perf_counter() - t0) / k
    same = bool((cpu_masks == g_masks.astype(np.uint32)).all()) and bool((cpu_chunks == g_chunks.reshape(k, -1)).all())
    total = int(max(threads, 0.6 * budget_s / per * threads))
    done = [0] * threads

    def worker(tid):
        buf = np.zeros(7500, np.uint8)
        for i in range(tid, total, threads):
            one(caps[i % k], buf)
            done[tid] += 1

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(sum(done) / dt, 2), "unit": "captures/s", "cores": ncores, "threads": threads, "kind": "reference", "opencv": "cv-shim (scalar)",
            "sample": f"{sum(done)} extract + decode runs over {k} of the {w}x{h} captures, {threads} threads, {dt:.1f} s wall",
            "single_thread_captures_per_s": round(1.0 / per, 1), "chunks_equal_gpu": same}


def stream_ms(dec, inputs, outs, steps, warmup, pipelined, stream, dev, pre=False):
    """ms per step of a continuous stream of batches (rotating through `inputs`), N = 1, no exchange."""
    D = dec.pipeline_depth if pipelined else 1

    def go(k):
        fr = inputs[k % len(inputs)]
        chunks, masks = outs[k % len(outs)]
        if pipelined:
            dec.decode_batch_pipelined(fr.data_ptr(), fr.shape[0], chunks.data_ptr(), masks.data_ptr(), pre, 2, stream.cuda_stream)
        else:
            dec.decode_batch_device(fr.data_ptr(), fr.shape[0], chunks.data_ptr(), masks.data_ptr(), pre, 2, stream.cuda_stream)

    assert len(outs) >= D
    for k in range(warmup):
        go(k)
    if pipelined:
        dec.pipeline_wait(stream.cuda_stream, 0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        go(warmup + k)
    if pipelined:
        dec.pipeline_wait(stream.cuda_stream, 0)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def stage_times(dec, frames, outs, stream, dev, reps=5, pre=False):
    """per-kernel ms: HIP events on the launch stream, recorded inside the library (un-split chain, one launch per kernel)"""
    acc = {}
    dec.enable_timing(True)
    chunks, masks = outs
    for rep in range(reps + 1):
        dec.decode_batch_device(frames.data_ptr(), frames.shape[0], chunks.data_ptr(), masks.data_ptr(), pre, 2, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        if rep == 0:
            continue          # the first ordinary call after the pipelined steps is not representative
        for k, v in dec.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    dec.enable_timing(False)
    return acc


def extras(dec, dev, stream, n, outs, steps):
    """N = 1 secondary rows (same process, same box, after the headline loop)."""
    out = {}
    synth = framegen.FrameSynth(dev)
    # ---- BASELINE configs[2]: n frames, 99 substituted cells each (rng 5678 + frame), every RS block in its error path
    try:
        payload = framegen.synth_payload(n, seed=1234, device=dev)
        tiles = framegen.inject_cell_errors(synth.cell_tiles(payload), n_errors=99, seed=5678)
        fe = torch.empty((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
        for lo in range(0, n, 64):
            synth.render(tiles[lo:lo + 64], out=fe[lo:lo + 64])
        del tiles
        ms_p = stream_ms(dec, [fe], outs, steps, 4, True, stream, dev)
        ok = all(bool((m == 0xFFF).all().item()) and bool((c == payload).all().item()) for c, m in outs[:dec.pipeline_depth])
        ms_u = stream_ms(dec, [fe], outs, max(8, steps // 4), 2, False, stream, dev)
        st = stage_times(dec, fe, outs[0], stream, dev, reps=3)
        out["config3_cell_errors"] = {"frames": n, "substituted_cells_per_frame": 99, "ms_per_step": round(ms_p, 4),
                                      "frames_per_s": round(n / ms_p * 1e3, 1), "no_pipeline_ms_per_step": round(ms_u, 4),
                                      "payload_ok": ok, "stage_ms": {k: round(v, 4) for k, v in st.items()}}
        del fe
    except Exception as e:   # a secondary row must never take the headline down
        out["config3_cell_errors"] = {"error": repr(e)}
    # ---- host-fed decode: pinned host frames in, chunks back to the host, H2D + D2H inside the timed region
    try:
        m = min(n, 256)
        payload, fr = make_frames(m, dev, 4242, dec, check=False)
        host = fr.cpu().pin_memory()
        del fr
        hv = host.numpy()
        dec.decode_batch(hv[:8])
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            total, chunks, masks = dec.decode_batch(hv)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        ok = total == m * 7500 and bool((torch.from_numpy(chunks.reshape(m, -1)) == payload.cpu()).all())
        out["host_fed"] = {"frames": m, "ms": round(best * 1e3, 3), "frames_per_s": round(m / best, 1),
                           "pcie_GBs": round(m * modeb.FRAME_RGB_BYTES / best / 1e9, 2), "payload_ok": ok,
                           "note": "pinned host memory -> cimbar_hip_decode_batch(host in, host out): one H2D copy + decode + D2H, synchronous"}
        # ---- one frame per call, the shape of the reference's own decode loop (cimbar.cpp:124-171: Decoder::decode_fountain per image): a host
        # frame in, its chunks out, synchronous -- latency, not throughput; next to the reference's single-thread figure of cpu_baseline
        try:
            k = 64
            for q in range(8):
                dec.decode_frame(hv[q])
            t0 = time.perf_counter()
            good = 0
            for q in range(k):
                r, c1, m1 = dec.decode_frame(hv[q])
                good += int(r)
            dt = (time.perf_counter() - t0) / k
            out["single_frame"] = {"frames": k, "ms_per_frame": round(dt * 1e3, 4), "frames_per_s": round(1.0 / dt, 1), "all_bytes_good": good == k * 7500,
                                   "note": "cimbar_hip_decode_frame: one pinned host frame in (H2D), eight kernels, chunks + mask back (D2H), one call "
                                           "per frame on one context -- what a drop-in Decoder::decode_fountain call costs"}
        except Exception as e:
            out["single_frame"] = {"error": repr(e)}
        # ---- the same loop with frames in flight (cimbar_hip_decode_frame_async / _wait): frame k+1's H2D copy beside frame k's kernels, chunks
        # taken in order -- what the adapter's Decoder::decode_fountain_overlapped does under cimbar.cpp:124-171's loop
        try:
            k, depth = 512, dec.pipeline_depth          # (a repetition must outlast the clock ramp-up after the synchronous loop above: ~20 ms)
            tickets = [dec.decode_frame_async(hv[q]) for q in range(depth)]
            for t in tickets:
                dec.decode_frame_wait(t)
            best, good = None, 0
            for _rep in range(3):
                tickets, good = [], 0
                t0 = time.perf_counter()
                for q in range(k):
                    tickets.append(dec.decode_frame_async(hv[q % m]))
                    if len(tickets) >= depth:
                        good += int(dec.decode_frame_wait(tickets.pop(0))[0])
                while tickets:
                    good += int(dec.decode_frame_wait(tickets.pop(0))[0])
                dt = (time.perf_counter() - t0) / k
                best = dt if best is None or dt < best else best
            out["single_frame_overlapped"] = {"frames": k, "in_flight": depth, "ms_per_frame": round(best * 1e3, 4), "frames_per_s": round(1.0 / best, 1),
                                              "all_bytes_good": good == k * 7500, "pcie_GBs": round(modeb.FRAME_RGB_BYTES / best / 1e9, 2),
                                              "note": "cimbar_hip_decode_frame_async + _wait on one context, pinned host frames, chunks taken in ticket order"}
        except Exception as e:
            out["single_frame_overlapped"] = {"error": repr(e)}
        del host, hv
    except Exception as e:
        out["host_fed"] = {"error": repr(e)}
    # ---- the ingest library: host threads stage / PNG-decode into a pinned ring, H2D on a copy stream overlapped with the decode
    try:
        import tempfile
        from PIL import Image
        from libcimbar_amd import ingest
        m = 512
        payload, fr = make_frames(128, dev, 5151, dec, check=False)
        host128 = fr.cpu().numpy()
        del fr
        host = np.ascontiguousarray(np.tile(host128, (m // 128, 1, 1, 1)))
        ing = ingest.Ingest(dec, threads=0, batch_frames=64, ring=3)
        ing.run_raw(host[:64], collect=False)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            total, _c, _m = ing.run_raw(host, collect=False)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        out["ingest_raw"] = {"frames": m, "ms": round(best * 1e3, 3), "frames_per_s": round(m / best, 1), "good_bytes_ok": total == m * 7500,
                             "pcie_GBs": round(m * modeb.FRAME_RGB_BYTES / best / 1e9, 2),
                             "note": "pageable host frames -> cimbar_ingest_run_raw: staging threads -> pinned ring (3 x 64 frames) -> H2D on a copy "
                                     "stream overlapped with cimbar_hip_decode_batch_pipelined -> D2H"}
        hp = torch.from_numpy(host).pin_memory()
        del host
        ing.run_raw_ptr(hp.data_ptr(), 64)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            total = ing.run_raw_ptr(hp.data_ptr(), m)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        out["ingest_pinned"] = {"frames": m, "ms": round(best * 1e3, 3), "frames_per_s": round(m / best, 1), "good_bytes_ok": total == m * 7500,
                                "pcie_GBs": round(m * modeb.FRAME_RGB_BYTES / best / 1e9, 2),
                                "note": "page-locked host frames -> cimbar_ingest_run_raw: no staging, H2D straight from the caller's buffer on the copy "
                                        "stream, overlapped with the pipelined decode and the D2H of the chunks"}
        del hp
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for k in range(128):
                pth = os.path.join(td, f"f{k:03d}.png")
                Image.fromarray(host128[k]).save(pth, compress_level=1)
                paths.append(pth)
            size = sum(os.path.getsize(x) for x in paths) / len(paths)
            paths = paths * (m // 128)
            ing.run_files(paths[:64])
            t0 = time.perf_counter()
            total, chunks, masks = ing.run_files(paths)
            dt = time.perf_counter() - t0
            tm = ing.timings()
        ok = total == m * 7500 and bool((torch.from_numpy(chunks[:128]) == payload.cpu()).all())
        out["ingest_png"] = {"files": m, "ms": round(dt * 1e3, 3), "frames_per_s": round(m / dt, 1), "payload_ok": ok, "avg_png_bytes": int(size),
                             "host_decode_cpu_s": round(tm["host_fill_s"], 3), "device_wait_s": round(tm["device_wait_s"], 4),
                             "note": "PNG files -> cimbar_ingest_run_files (read + inflate + un-filter on a host thread pool) -> the same ring"}
        ing.close()
        # ---- the same files with the PNGs decoded ON THE DEVICE (cimbar_ingest_create_ex, CIMBAR_INGEST_PNG_DEVICE): the host threads only
        # read the files and move their IDAT bytes; inflate + un-filter are kernels (one wavefront per image: large batches fill the GPU)
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for k in range(128):
                pth = os.path.join(td, f"f{k:03d}.png")
                Image.fromarray(host128[k]).save(pth, compress_level=1)
                paths.append(pth)
            mm = 32768
            paths = paths * (mm // 128)
            ing = ingest.Ingest(dec, threads=0, batch_frames=2048, ring=3, png_device=True, zbytes_per_frame=400000)
            ing.run_files(paths[:256])
            t0 = time.perf_counter()
            total, chunks, masks = ing.run_files(paths)
            dt = time.perf_counter() - t0
            tm, ps = ing.timings(), ing.png_stats()
            ing.close()
            ok = total == mm * 7500 and bool((torch.from_numpy(chunks[:128]) == payload.cpu()).all()) and bool((torch.from_numpy(chunks[-128:]) == payload.cpu()).all())
            out["ingest_png_device"] = {"files": mm, "ms": round(dt * 1e3, 3), "frames_per_s": round(mm / dt, 1), "payload_ok": ok,
                                        "host_cpu_s": round(tm["host_fill_s"], 3), "device_wait_s": round(tm["device_wait_s"], 4),
                                        "pcie_bytes_per_frame": int(ps["bytes_to_device"] / mm), "refused": ps["refused_by_host_walk"] + ps["refused_by_device"],
                                        "note": "PNG files -> cimbar_ingest_run_files in device PNG mode: compressed bytes over PCIe, k_png_inflate + "
                                                "k_png_unfilter + decode on the device (up to 3 batches of 2048 in flight: the one-stream inflate with the all-offsets turn; measured against 3 x 4096 with the device-chosen kernel: 26.7 vs 24.7 k frames/s)"}
            # the two PNG kernels alone on device-resident streams, for the two kinds of PNG a frame comes as: Pillow's writer (adaptive
            # filters, deflate level 1: long matches) and cv::imwrite's defaults, i.e. the reference encoder's own files (Sub filter on every
            # row, Z_RLE, level 1: mostly 2-bit literals -- five times the tokens)
            from libcimbar_amd import decoder as _d
            import ctypes as _ct
            import struct as _st, zlib as _zl
            L = _d.load_library()

            def cv_default_zlib(img):
                rows = img.reshape(img.shape[0], -1).astype(np.int16)
                sub = rows.copy()
                sub[:, 3:] -= rows[:, :-3]
                raw = np.concatenate([np.ones((img.shape[0], 1), np.uint8), (sub & 0xFF).astype(np.uint8)], axis=1).tobytes()
                co = _zl.compressobj(1, _zl.DEFLATED, 15, 8, _zl.Z_RLE)
                return co.compress(raw) + co.flush()

            def png_kernels(zstreams, npng, what):
                desc = (_d.PngDesc * npng)()
                blob, offs = bytearray(), []
                for z in zstreams:
                    while len(blob) % 16:
                        blob.append(0)
                    offs.append((len(blob), len(z)))
                    blob += z
                while len(blob) % 16:
                    blob.append(0)
                for i in range(npng):
                    desc[i].zoff, desc[i].zlen = offs[i % len(offs)]
                    desc[i].width, desc[i].height, desc[i].color_type = modeb.IMG, modeb.IMG, 2
                d_z = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).to(dev)
                d_desc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
                ss = int(L.cimbar_hip_png_scratch_bytes(modeb.IMG, modeb.IMG, 2))
                d_scr = torch.empty(npng * ss, dtype=torch.uint8, device=dev)
                d_rgb = torch.empty((npng, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
                d_st = torch.zeros(npng, dtype=torch.int32, device=dev)
                best = None
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    L.cimbar_hip_png_decode_batch(dev.index, d_z.data_ptr(), d_z.numel(), d_desc.data_ptr(), npng, d_scr.data_ptr(), ss, d_rgb.data_ptr(),
                                                  modeb.FRAME_RGB_BYTES, d_st.data_ptr(), _ct.c_void_p(stream.cuda_stream))
                    torch.cuda.synchronize(dev)
                    dtk = time.perf_counter() - t0
                    best = dtk if best is None or dtk < best else best
                fr128 = torch.from_numpy(host128).to(dev)
                okk = bool((d_st == 0).all().item()) and all(bool((d_rgb[i] == fr128[i % len(offs)]).all().item()) for i in range(0, npng, 61))
                del d_scr, d_rgb, fr128, d_z
                torch.cuda.empty_cache()
                return {"images": npng, "ms": round(best * 1e3, 2), "images_per_s": round(npng / best, 1), "pixels_ok": okk,
                        "avg_zlib_bytes": int(sum(l for _o, l in offs) / len(offs)),
                        # input = the zlib streams, output = the RGB frames; the inflate kernels are bound by the CU's one scalar unit
                        "roofline": {"bound": "salu", "kernel": "k_png_inflate* (device-chosen) + k_png_unfilter", "unit": "GB/s", "peak": 8000.0,
                                     "achieved": round((sum(l for _o, l in offs) / len(offs) + 2 * modeb.FRAME_RGB_BYTES + modeb.IMG) * npng / best / 1e9, 2),
                                     "frac": round((sum(l for _o, l in offs) / len(offs) + 2 * modeb.FRAME_RGB_BYTES + modeb.IMG) * npng / best / 8e12, 5),
                                     "basis": "zlib bytes in + filtered scanlines out and in again + RGB out, both kernels' wall time", "traffic": None},
                        "note": f"cimbar_hip_png_decode_batch on device-resident zlib streams of 1024x1024 frame PNGs ({what}): inflate (the device "
                                "picks the kernel from the first stream's first block at this size: four streams per wavefront for long matches, the "
                                "one-stream all-offsets turn for short literal codes) + un-filter"}

            # files -> chunks once more with the reference encoder's kind of file
            def cv_default_png(img):
                def chunk(t, d):
                    return _st.pack(">I", len(d)) + t + d + _st.pack(">I", _zl.crc32(t + d) & 0xFFFFFFFF)
                return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", _st.pack(">IIBBBBB", img.shape[1], img.shape[0], 8, 2, 0, 0, 0)) +
                        chunk(b"IDAT", cv_default_zlib(img)) + chunk(b"IEND", b""))
            try:
                with tempfile.TemporaryDirectory() as td2:
                    paths2 = []
                    for k in range(128):
                        pth = os.path.join(td2, f"c{k:03d}.png")
                        with open(pth, "wb") as f:
                            f.write(cv_default_png(host128[k]))
                        paths2.append(pth)
                    size2 = sum(os.path.getsize(x) for x in paths2) / len(paths2)
                    mm2 = 16384
                    paths2 = paths2 * (mm2 // 128)
                    ing = ingest.Ingest(dec, threads=0, batch_frames=2048, ring=3, png_device=True, zbytes_per_frame=460000)
                    ing.run_files(paths2[:256])
                    t0 = time.perf_counter()
                    total, chunks, masks = ing.run_files(paths2)
                    dt = time.perf_counter() - t0
                    ps = ing.png_stats()
                    ing.close()
                    ok2 = total == mm2 * 7500 and bool((torch.from_numpy(chunks[:128]) == payload.cpu()).all())
                    out["ingest_png_device_cv_writer"] = {"files": mm2, "ms": round(dt * 1e3, 3), "frames_per_s": round(mm2 / dt, 1), "payload_ok": ok2,
                                                          "avg_png_bytes": int(size2), "refused": ps["refused_by_host_walk"] + ps["refused_by_device"],
                                                          "note": "the same path on PNGs as cv::imwrite writes them by default (Sub filter, Z_RLE, level 1: the "
                                                                  "reference encoder's files) -- five times the deflate tokens of Pillow's"}
            except Exception as e:
                out["ingest_png_device_cv_writer"] = {"error": repr(e)}
            out["png_device_kernels"] = png_kernels([_d.png_split(open(paths[k], "rb").read())[5] for k in range(128)], 8192, "Pillow, compress_level 1")
            out["png_device_kernels_cv_writer"] = png_kernels([cv_default_zlib(host128[k]) for k in range(32)], 8192,
                                                              "cv::imwrite's defaults = the reference encoder's files: Sub filter, Z_RLE, level 1")
    except Exception as e:
        out["ingest"] = {"error": repr(e)}
    # ---- modes 67 ("Bm", Conf8x8_mini: 1024x720 frames, 12 x 429 bytes), 66 ("Bu", Conf8x8_micro: 736x637, 6 x 540) and 4 (legacy 4-colour:
    # mode B's grid, one coupled Reed-Solomon stream, 10 x 750; 8 = the 8-colour one, 10 x 875): the same kernels compiled for the other configurations, each with its own context
    for other in (67, 66, 4, 8):
      try:
        from libcimbar_amd import HipDecoder, geometry
        g = geometry.for_mode(other)
        d67 = HipDecoder(dec.device, other)
        payload = framegen.synth_payload(n, seed=6767, device=dev, mode=other)
        f67 = torch.empty((n, *g.FRAME_SHAPE), dtype=torch.uint8, device=dev)
        d67.encode_batch_device(payload.data_ptr(), n, f67.data_ptr())
        o67 = [(torch.zeros((n, g.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(d67.pipeline_depth)]
        ms_p = stream_ms(d67, [f67], o67, steps, 4, True, stream, dev)
        ok = all(bool((m == g.FULL_MASK).all().item()) and bool((c == payload).all().item()) for c, m in o67)
        st = stage_times(d67, f67, o67[0], stream, dev, reps=3)
        alg = n * (g.FRAME_RGB_BYTES + g.FRAME_BYTES + 4)
        out["mode%d" % other] = {"frames": n, "frame": "%dx%d" % (g.IMG_W, g.IMG_H), "ms_per_step": round(ms_p, 4), "frames_per_s": round(n / ms_p * 1e3, 1),
                                 "payload_ok": ok, "whole_path_hbm_frac": round(alg / (ms_p * 1e-3) / 8e12, 4),
                                 "threshold_hbm_frac": round(alg / (st["threshold"] * 1e-3) / 8e12, 4), "stage_ms": {k: round(v, 4) for k, v in st.items()}}
        del f67, o67
        d67.close()
      except Exception as e:
        out["mode%d" % other] = {"error": repr(e)}
    # ---- BASELINE configs[3] at N = 1: the 8192-frame fountain stream of a 16 MiB file, slabs of 1024, the reference's sink fed inside the
    # timed region by a host thread while the next slab decodes (bench.py --config 4 is the same code under torchrun)
    try:
        import bench_config4 as config4
        torch.cuda.empty_cache()
        c4 = config4.bench(dec, dev, 0, 1, argparse.Namespace(frames=1024, steps=2, warmup=1))
        out["config4_n1"] = {k: c4[k] for k in ("value", "unit", "ms_per_step", "stage_s", "decode_only_frames_per_s", "end_to_end_over_decode_only", "without_solve", "sink",
                                                "chunks_match_encoded_stream")}
        out["config4_n1"]["workload"] = c4["config"]["workload"]
        torch.cuda.empty_cache()
    except Exception as e:
        out["config4_n1"] = {"error": repr(e)}
    try:
        from tools import extractbench
        sample = {}
        out.update(extractbench.run(dec, dev, stream, synth, sample=sample))
        torch.cuda.empty_cache()
        try:
            out["config5_extract"]["cpu_baseline"] = cpu_baseline_config5(sample)
        except Exception as e:
            out["config5_extract"]["cpu_baseline"] = {"error": repr(e)}
        del sample
        # the same at 1024 captures per batch: the exact flood keeps one (two-wavefront) workgroup per frame busy for ~35 ms whatever the batch,
        # so the per-batch time barely moves and throughput follows the batch size until every CU holds its four frames
        out.update(extractbench.run(dec, dev, stream, synth, n=1024, reps=2, key="config5_extract_1024"))
        torch.cuda.empty_cache()
        # the same captures as the camera hands them over (web/recv-worker.js: VideoFrames in NV12): `format` 12 of the reference's C ABI, converted
        # inside the kernels that read the capture -- and host-fed, where the format decides how many bytes cross PCIe
        try:
            out.update(extractbench.run(dec, dev, stream, synth, n=1024, reps=2, key="config5_extract_nv12", fmt=12))
            torch.cuda.empty_cache()
            out.update(extractbench.run_host_fed(dec, dev, stream, n=256))
            torch.cuda.empty_cache()
        except Exception as e:
            out["config5_extract_nv12"] = {"error": repr(e)}
        # larger batches: the replay's residency (frames per CU) is what bounds a batch's time. Above 1024 frames per launch the
        # library runs the replay's dense instance (eight frames per CU: k_flood3<4095, true>)
        for big in (2048, 4096):
            try:
                out.update(extractbench.run(dec, dev, stream, synth, n=big, reps=2, key=f"config5_extract_{big}"))
            except Exception as e:
                out[f"config5_extract_{big}"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        # and as a stream of batches through two / three contexts (the reference CLI's worker threads): extract and threshold of one batch run
        # beside the flood replay of another
        try:
            out.update(extractbench.run_stream(dev, n=1024, contexts=2, batches=6, key="config5_stream_1024"))
            torch.cuda.empty_cache()
            out.update(extractbench.run_stream(dev, n=256, contexts=3, batches=12, key="config5_stream_256"))
            torch.cuda.empty_cache()
            out.update(extractbench.run_stream(dev, n=2048, contexts=2, batches=6, key="config5_stream_2048"))
            torch.cuda.empty_cache()
            out.update(extractbench.run_stream(dev, n=4096, contexts=2, batches=4, key="config5_stream_4096"))
            torch.cuda.empty_cache()
        except Exception as e:
            out["config5_stream_1024"] = {"error": repr(e)}
    except ImportError:
        pass
    except Exception as e:
        out["config5_extract"] = {"error": repr(e)}
    # every row that names what bounds it gets a `roofline` object in the headline's shape (host-fed rows: the PCIe link, 64 GB/s for the
    # Gen5 x16 of these boxes; device-PNG ingest: the inflate kernels, scalar-unit bound -- their HBM figure is stated so that the rows read alike)
    PCIE_GBS = 64.0
    for key in ("host_fed", "ingest_raw", "ingest_pinned"):
        row = out.get(key)
        if isinstance(row, dict) and "pcie_GBs" in row and "roofline" not in row:
            row["roofline"] = {"bound": "pcie", "kernel": None, "achieved": row["pcie_GBs"], "peak": PCIE_GBS, "unit": "GB/s",
                               "frac": round(row["pcie_GBs"] / PCIE_GBS, 4), "basis": "frame bytes host -> device per second of the whole call", "traffic": None}
    for key in ("ingest_png_device", "ingest_png_device_cv_writer"):
        row = out.get(key)
        if isinstance(row, dict) and "frames_per_s" in row and "roofline" not in row:
            algo = (row.get("pcie_bytes_per_frame") or row.get("avg_png_bytes") or 0) + 2 * (modeb.FRAME_RGB_BYTES + modeb.IMG) + modeb.FRAME_RGB_BYTES
            gbs = algo * row["frames_per_s"] / 1e9
            row["roofline"] = {"bound": "salu", "kernel": "k_png_inflate* (device-chosen) + k_png_unfilter, then the decode chain", "achieved": round(gbs, 2), "peak": 8000.0,
                               "unit": "GB/s", "frac": round(gbs / 8000.0, 5),
                               "basis": "compressed bytes in + filtered scanlines out and in again + RGB out + RGB into the decoder, per second of the whole run", "traffic": None}
    return out


def summary(line):
    """The secondary rows in one small object, printed as the LAST key of the JSON line so that a tail of the output still holds them."""
    ex = line.get("extra") or {}

    def get(key, field):
        row = ex.get(key)
        return row.get(field) if isinstance(row, dict) else None
    c4 = ex.get("config4_n1") if isinstance(ex.get("config4_n1"), dict) else {}
    ws = c4.get("without_solve")
    return {
        "configs1_ms_per_step": line.get("ms_per_step"), "configs1_k1_frac": (line.get("roofline") or {}).get("frac"),
        "configs1_whole_path_frac": (line.get("roofline") or {}).get("whole_path_frac"),
        "configs2_ms_per_step": get("config3_cell_errors", "ms_per_step"),
        "configs3_n1_frames_per_s": c4.get("value"), "configs3_n1_without_solve": ws.get("frames_per_s") if isinstance(ws, dict) else ws,
        "configs4_captures_per_s": {"256": get("config5_extract", "captures_per_s"), "1024": get("config5_extract_1024", "captures_per_s"),
                                    "1024_nv12": get("config5_extract_nv12", "captures_per_s"), "2048": get("config5_extract_2048", "captures_per_s"),
                                    "4096": get("config5_extract_4096", "captures_per_s"), "stream_2048": get("config5_stream_2048", "captures_per_s"),
                                    "stream_4096": get("config5_stream_4096", "captures_per_s")},
        "png_files_to_chunks_frames_per_s": {"pillow": get("ingest_png_device", "frames_per_s"), "cv_writer": get("ingest_png_device_cv_writer", "frames_per_s")},
        "png_kernels_images_per_s": {"pillow": get("png_device_kernels", "images_per_s"), "cv_writer": get("png_device_kernels_cv_writer", "images_per_s")},
        "single_frame_ms": get("single_frame", "ms_per_frame"), "single_frame_overlapped_frames_per_s": get("single_frame_overlapped", "frames_per_s"),
        "host_fed_frames_per_s": get("host_fed", "frames_per_s"),
        "mode_k1_frac": {str(m): get("mode%d" % m, "threshold_hbm_frac") for m in (67, 66, 4, 8)},
        "cpu_baseline_frames_per_s": (line.get("cpu_baseline") or {}).get("value"),
    }


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs this process could open, WITHOUT creating a HIP context in the parent (the ranks it starts must be the first to touch their device)."""
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def self_launch(n, argv, check_gpus=True):
    """`python bench.py --gpus N` without a launcher around it: re-run this file as N ranks under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1), which is the command the driver itself uses for N > 1. Returns the exit code of the job. With fewer than N
    GPUs visible nothing is started: a 1-GPU number must never be printed for an N-GPU request."""
    if check_gpus:
        have = visible_gpus()
        if have < n:
            print(f"bench: needs {n} GPUs, found {have}", file=sys.stderr, flush=True)
            return 3
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    print("bench: starting " + " ".join(cmd[1:]), file=sys.stderr, flush=True)
    sys.stdout.flush()
    return subprocess.run(cmd, env=env, stdout=sys.stdout).returncode          # (the ranks write their line to THIS process's real stdout: claim_stdout)


def launch_check(rank, local_rank, world):
    """--launch-check: the ranks exist, know who they are and can reach each other -- over gloo, no GPU touched. Rank 0 prints one JSON line."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    me = {"rank": rank, "local_rank": local_rank, "world_size": world, "pid": os.getpid(),
          "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}}
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        ok = int(t.item()) == world * (world + 1) // 2 and sorted(e["rank"] for e in everyone) == list(range(world)) \
            and sorted(e["local_rank"] for e in everyone) == list(range(world))
        dist.destroy_process_group()
    else:
        everyone, ok = [me], True
    if rank == 0:
        print(json.dumps({"launch_check": "ok" if ok else "FAILED", "n_ranks": world, "ranks": everyone}), flush=True)
    return 0 if ok else 4


def claim_stdout():
    """The JSON line must be the only thing on stdout: libraries print there too (RCCL's version banner lands in the C runtime's buffer and is flushed
    at exit, BEHIND the line). File descriptor 1 is pointed at stderr for everything else; Python's sys.stdout keeps the real one."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = real


def main():
    claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step (BASELINE configs[1]: 1024)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4), help="2: BASELINE configs[1] (default); 4: configs[3], the sharded fountain stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the library's own exchange (cimbar_hip_gather_chunks) cannot be set up, fall back to torch.distributed.gather "
                         "instead of failing -- the line then says so in config.exchange, and is NOT a measurement of this library's exchange")
    ap.add_argument("--reps", type=int, default=0, help="repeat the timed region this many times and report the median (0: 5 when --steps < 100, else 1)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one ordinary call per step instead of the pipelined entry point (kernels of different steps never overlap: "
                         "what tools/gpu_profile.sh uses so that every traced dispatch is one kernel running alone)")
    ap.add_argument("--probe-run", action="store_true",
                    help="timing experiment with a probe build of the library (-DCIMBAR_PROBES, CIMBAR_HIP_LIB=libcimbar_amd/variants/libcimbar_hip_probes.so, "
                         "tools/skip_probe.sh): prints the timing fields only -- no metric / value, such a line is not a result")
    ap.add_argument("--no-solo-exchange", dest="solo_exchange", action="store_false",
                    help="N = 1 only: do NOT run every step's outputs through the library's exchange with a one-rank RCCL communicator inside the timed loop "
                         "(the default does what each rank of an N-GPU job does: +0.4 ... 1.6 % on the step, profiles/r06m_bench_exchange_*); the line then "
                         "checks the exchange once outside the timed region instead (config.exchange_selfcheck)")
    ap.add_argument("--solo-exchange", dest="solo_exchange", action="store_true", help="(the default)")
    ap.set_defaults(solo_exchange=True)
    ap.add_argument("--exchange-side-stream", action="store_true",
                    help="issue the exchange on a stream of its own once the step it reads is complete (rounds 3-5) instead of behind the step on the step's own "
                         "pipeline stream (cimbar_hip_pipeline_gather): for A/B runs")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher self-test, touches no GPU: start the --gpus ranks exactly as a measurement would, rendezvous them over gloo on "
                         "127.0.0.1, and print one JSON line with every rank's RANK / LOCAL_RANK / WORLD_SIZE")
    args = ap.parse_args()
    if os.environ.get("CIMBAR_HIP_DEBUG_SKIP", "0") not in ("", "0") and not args.probe_run:
        # (the product library ignores the variable -- the switch only exists in a -DCIMBAR_PROBES build -- but a line printed while it is set would
        # invite exactly the doubt it is easiest to avoid)
        raise SystemExit("bench: CIMBAR_HIP_DEBUG_SKIP is set (chain kernels dropped, results wrong on purpose): no measurement line is printed "
                         "under it; timing experiments go through --probe-run (tools/skip_probe.sh)")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` (the shape of the driver's N = 1 command) IS the N-rank job: one process per GPU, started here
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], check_gpus=not args.launch_check))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never an n_gpus line for another N than the one asked for
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}: start exactly one rank per requested GPU "
                         f"(`python bench.py --gpus {args.gpus}` does that by itself)")
    if args.launch_check:
        raise SystemExit(launch_check(rank, local_rank, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    if torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench: needs {world} GPUs, found {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the decoder's streams are created first, right behind the null stream: how streams fall onto the runtime's hardware queues
    # depends on creation order, and this is the order the pipeline was measured in (DESIGN.md "Launch structure")
    dec = HipDecoder(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.config == 4:
        import bench_config4 as config4
        line = config4.bench(dec, dev, rank, world, args)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    n = args.frames
    # The batches form a continuous stream, so the library's pipelined entry point is used: up to D = pipeline_depth steps are in flight
    # on the context's own streams, the threshold pass of one overlapping the short kernels of the others (colour-correction carry-over
    # still in batch order). A step's outputs are consumed D-1 steps later: with N > 1 the RCCL gather of step k-D+1 is issued right
    # after step k has been enqueued. D output buffer sets (and, on rank 0, D gather destinations).
    D = 1 if args.no_pipeline else dec.pipeline_depth
    # output buffer sets: one per step in flight (the exchange of a step rides on the step's own pipeline stream, so a set always meets the same stream)
    NB = max(D, 2)
    # R distinct input batches, decoded in rotation: steps in flight at the same time never read the same frames (two threshold passes
    # walking the same addresses a few frames apart would share lines through L2 / the Infinity Cache, which a real stream cannot)
    R = max(dec.pipeline_depth, 4)          # (12 GB per rotation: nothing survives in the 256 MiB Infinity Cache from one use to the next)
    batches = [make_frames(n, dev, seed=1234 + 97 * rank + 7919 * b, dec=dec, check=(b == 0)) for b in range(R)]
    payloads = [p for p, _ in batches]
    inputs = [f for _, f in batches]
    outs = [(torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros((n,), dtype=torch.int32, device=dev))
            for _ in range(NB)]
    gathered = [(torch.zeros((world * n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev),
                 torch.zeros((world * n,), dtype=torch.int32, device=dev)) if rank == 0 else None for _ in range(NB)]
    stream = torch.cuda.current_stream(dev)
    which = [0] * NB          # input batch last decoded into output set b

    def issue(b, k):
        chunks, masks = outs[b]
        fr = inputs[k % R]
        which[b] = k % R
        if args.no_pipeline:
            dec.decode_batch_device(fr.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, stream.cuda_stream)
        else:
            dec.decode_batch_pipelined(fr.data_ptr(), n, chunks.data_ptr(), masks.data_ptr(), False, 2, stream.cuda_stream)

    def ready(keep_newest):
        if not args.no_pipeline:
            dec.pipeline_wait(stream.cuda_stream, keep_newest=keep_newest)

    # the exchange is the library's own (cimbar_hip_gather_chunks: ncclGather over RCCL / xGMI issued by libcimbar_hip.so); torch.distributed
    # is the rendezvous (it carries the communicator id) and the closing barrier, nothing on the data path
    exchange, exchange_name = None, None
    if world > 1:
        try:
            exchange = multigpu.LibraryGather(dec, dev)
            exchange_name = "cimbar_hip_gather_chunks (RCCL ncclGather issued by the library)"
        except Exception as e:          # e.g. no librccl next to this torch build
            exchange, exchange_name = None, f"torch.distributed.gather (library exchange unavailable: {e!r})"
        flags = torch.tensor([1 if exchange is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)          # every rank takes the same path
        if int(flags.item()) == 0:
            if exchange is not None:
                exchange.close()
                exchange, exchange_name = None, "torch.distributed.gather (library exchange unavailable on another rank)"
            # a scaling line must measure the library's exchange or nothing: torch's gather only on explicit request
            if not args.allow_fallback:
                raise SystemExit(f"bench: cimbar_hip_gather_chunks could not be set up on every rank ({exchange_name}); "
                                 "pass --allow-fallback to time torch.distributed.gather instead")
    if world == 1 and not args.no_pipeline and args.solo_exchange:
        # N = 1: the steps go through the library's exchange with a communicator of ONE rank, exactly as the ranks of an N-GPU job do: the record the driver
        # takes then shows librccl bound and ncclGather issued inside the timed loop (config.exchange_ranks = what ncclCommCount says). Since the exchange
        # rides on the step's own pipeline stream (cimbar_hip_pipeline_gather) this costs the step 0.4 ... 1.6 % (profiles/r06m_bench_exchange_*: 0.716 /
        # 0.719 and 0.722 / 0.733 ms, without / with, alternated); on a stream of its own it cost 10 % (--exchange-side-stream). --no-solo-exchange leaves
        # it out and checks the RCCL path with ONE gather outside the timed region instead (config.exchange_selfcheck). If RCCL cannot be loaded the line
        # says so and the steps run without it.
        # (set up on a helper thread with a deadline: a communicator that cannot be built -- no usable socket interface for RCCL's bootstrap in some sandbox --
        # must cost the line its exchange, never the line itself)
        box = {}

        def setup():
            try:
                torch.cuda.set_device(local_rank)
                box["exchange"] = multigpu.LibraryGather(dec, dev, copy_only=os.environ.get("CIMBAR_BENCH_SOLO_EXCHANGE") == "copy")
            except Exception as e:
                box["error"] = e
        th = threading.Thread(target=setup, daemon=True)
        th.start()
        th.join(float(os.environ.get("CIMBAR_BENCH_EXCHANGE_SETUP_S", "120")))
        if "exchange" in box:
            exchange = box["exchange"]
            exchange_name = "cimbar_hip_gather_chunks (RCCL ncclGather issued by the library; one-rank communicator)" if not exchange.copy_only \
                else "a device-to-device copy in the exchange's place (experiment: the exchange's stream / event structure without RCCL's kernel)"
        else:
            why = repr(box["error"]) if "error" in box else "ncclCommInitRank did not return within the deadline"
            exchange, exchange_name = None, f"none (N = 1; the library's RCCL exchange could not be set up: {why})"
    # The exchange of a step is issued right behind it on the step's own pipeline stream (cimbar_hip_pipeline_gather): no side stream, no events; the
    # pipeline wait that makes a step's outputs safe to read covers its gathered chunks as well. (--exchange-side-stream: the round-3..5 arrangement --
    # the gather of step k-D+1 on a stream of the exchange's own once that step is complete -- kept for A/B runs.)
    inline = None
    ready_gather = None
    if exchange is not None and not args.no_pipeline and not getattr(exchange, "copy_only", False) and not args.exchange_side_stream:
        inline = exchange.inline
    elif exchange is not None and not args.no_pipeline:
        def ready_gather(keep_newest):          # the exchange's stream (not the caller's) waits for the steps whose outputs it is about to read
            dec.pipeline_wait(exchange.stream.cuda_stream, keep_newest=keep_newest)
    if inline is not None:
        exchange_name = "cimbar_hip_pipeline_gather (RCCL ncclGather issued by the library behind each step on the step's own pipeline stream" + \
                        ("; one-rank communicator)" if world == 1 else ")")
    pipe = multigpu.StepPipeline(outs, D, issue, ready, gathered=gathered if exchange is not None or world > 1 else None, dst=0, gather=exchange,
                                 ready_for_gather=ready_gather, inline_gather=inline)
    step, drain = pipe.step, pipe.drain

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    # The timed region: EXACTLY --steps steps between two barriers (max over ranks). With few steps it lasts a few milliseconds, and boxes /
    # clocks wander by 5-10 % over such a span: the region is then repeated and the MEDIAN repetition is the one reported (all of them listed).
    nreps = args.reps if args.reps > 0 else (5 if args.steps < 100 else 1)
    rep_s = []
    for _ in range(nreps):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        barrier()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        rep_s.append(e)
    elapsed = sorted(rep_s)[len(rep_s) // 2]
    all_chunks, all_masks = pipe.last if pipe.last is not None else outs[(pipe.steps - 1) % NB]

    # correctness of what was just timed (outside the timed region): bit-exact payload, every chunk delivered, every buffer set
    ok = True
    for b, (chunks, masks) in enumerate(outs[:min(NB, pipe.steps)]):
        ok = ok and bool((masks == 0xFFF).all().item()) and bool((chunks == payloads[which[b]]).all().item())
    if pipe.exchanging and rank == 0:
        last_in = (pipe.steps - 1) % R
        ok = ok and all_chunks.shape[0] == world * n and bool((all_chunks[:n] == payloads[last_in]).all().item()) \
            and bool((all_masks == 0xFFF).all().item())
    if not ok and not args.probe_run:
        raise SystemExit("bench: decoded payload differs from what was encoded -- number would be meaningless")

    # N = 1 without an exchange in the loop: ONE gather of the last step's outputs through the library's RCCL path (a one-rank communicator), outside the
    # timed region, compared byte for byte -- the record then shows librccl bound and ncclGather issued by the library on this box
    def exchange_selfcheck():
        try:
            ex1 = multigpu.LibraryGather(dec, dev)
            c_in, m_in = outs[(pipe.steps - 1) % NB]
            c_out, m_out = torch.zeros_like(c_in), torch.zeros_like(m_in)
            ex1(c_in, m_in, dst=0, out=(c_out, m_out), async_op=False)
            torch.cuda.synchronize(dev)
            res = {"ranks": ex1.nranks, "gathered_equals_decoded": bool((c_out == c_in).all().item()) and bool((m_out == m_in).all().item()),
                   "what": "one cimbar_hip_gather_chunks (ncclGather over RCCL, one-rank communicator) of one step's chunks and masks, outside the timed region"}
            ex1.close()
            return res
        except Exception as e:
            return {"error": repr(e)}

    line = None
    if rank == 0:
        frames_per_s = world * n * args.steps / elapsed
        stage_acc = stage_times(dec, inputs[0], outs[0], stream, dev)
        dom = max(stage_acc, key=stage_acc.get)
        dom_ms = stage_acc[dom]
        # The pipelined loop launches the threshold kernel's TALL-strip instance for batches of 256 frames and more (fewer row steps and halo rows per
        # frame: better for the loop, a little slower as a launch of its own -- DESIGN.md K1); a call that runs alone, like the stage timing above,
        # takes the short-strip one. The roofline object describes the kernel the TIMED REGION ran: the tall instance, timed alone here through
        # a context that is told to use it everywhere; the short instance's figure stays next to it.
        short_ms, tall_used = dom_ms, False
        if dom == "threshold" and not args.no_pipeline and n >= 256 and os.environ.get("CIMBAR_HIP_K1_STRIPS", "auto") in ("auto", "tall"):
            prev_env = os.environ.get("CIMBAR_HIP_K1_STRIPS")
            try:
                os.environ["CIMBAR_HIP_K1_STRIPS"] = "tall"
                dec_t = HipDecoder(local_rank)
                tall_acc = stage_times(dec_t, inputs[0], outs[0], stream, dev)
                dec_t.close()
                stage_acc["threshold_tall_strips"] = tall_acc["threshold"]
                dom_ms, tall_used = tall_acc["threshold"], True
            finally:
                if prev_env is None:
                    os.environ.pop("CIMBAR_HIP_K1_STRIPS", None)
                else:
                    os.environ["CIMBAR_HIP_K1_STRIPS"] = prev_env
        achieved = ALGO_BYTES_PER_FRAME * n / (dom_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dom, n, tall=tall_used)
        trace_us, trace_src = trace_average(dom, tall=tall_used)
        line = {
            "metric": "decoded cimbar frames/s (1024x1024 mode-B)", "value": round(frames_per_s, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "timed_region": {"repetitions": nreps, "reported": "median", "ms_per_step_each": [round(e / args.steps * 1e3, 4) for e in rep_s],
                             "ms_per_step_min": round(min(rep_s) / args.steps * 1e3, 4), "ms_per_step_max": round(max(rep_s) / args.steps * 1e3, 4)},
            "config": {"workload": f"BASELINE configs[1]: {n} clean mode-B frames per GPU per step, device-resident, bit-exact",
                       "input": f"{R} distinct synthetic batches decoded in rotation (no step is fed from L2 / Infinity Cache)",
                       "frames_per_gpu_per_step": n, "distinct_input_batches": R,
                       "exchange": exchange_name,
                       "exchange_ranks": getattr(exchange, "nranks", None),          # ncclCommCount of the library's communicator (None: no exchange ran)
                       "exchanges_in_timed_run": pipe.gathers, "exchange_selfcheck": None,
                       "parallelism": f"frame-sharded x{world}" + (", RCCL gather to rank 0" if world > 1 else "") +
                                      ("" if args.no_pipeline else f"; {D} steps in flight (pipelined entry point)")},
            "roofline": {"bound": "hbm", "kernel": dom + (" (k_threshold<2, false, 7>: the tall-strip instance the timed loop launches, timed as a launch of its own)" if tall_used else ""),
                         "short_strip_instance_alone": {"ms": round(short_ms, 4), "frac": round(ALGO_BYTES_PER_FRAME * n / (short_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                        "note": "k_threshold<2, false, 2>: what a call that runs alone launches"} if tall_used else None,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes": ALGO_BYTES_PER_FRAME * n,
                         "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_src,
                         # the same kernel's average duration in the committed rocprofv3 kernel trace (another box of the pool, another day: boxes differ by 5-10 %)
                         "kernel_avg_us": round(dom_ms * 1e3, 2), "kernel_avg_us_trace": trace_us, "kernel_avg_us_trace_source": trace_src,
                         "whole_path_frac": round(frames_per_s / world * ALGO_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms": {k: round(v, 4) for k, v in stage_acc.items()},
        }
    if world == 1:
        if not args.no_pipeline:
            ms_u = stream_ms(dec, inputs, outs, args.steps, min(args.warmup, 4), False, stream, dev)
            line["no_pipeline"] = {"ms_per_step": round(ms_u, 4), "frames_per_s": round(n / ms_u * 1e3, 1),
                                   "whole_path_frac": round(n / ms_u * 1e3 * ALGO_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS, 5)}
        if not args.no_cpu_baseline:
            k = n          # every frame of the batch (round 5 compared 64 of them): ~6 s of one host thread at 1 024 frames
            chunks, masks = outs[0]
            dec.reset_ccm()
            dec.decode_batch_device(inputs[0].data_ptr(), k, chunks.data_ptr(), masks.data_ptr(), False, 2, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            line["cpu_baseline"] = cpu_baseline(inputs[0][:k].cpu().numpy(), chunks[:k].cpu().numpy())
            line["payload_sha_match"] = line["cpu_baseline"]["payload_sha_match"]
        if not args.no_extras:
            del batches
            inputs[1:] = []
            torch.cuda.empty_cache()
            line["extra"] = extras(dec, dev, stream, n, outs, max(20, args.steps // 4))
    if world == 1 and exchange is None and not args.solo_exchange and not args.probe_run:
        # (--no-solo-exchange; if the in-loop exchange was wanted and could not be set up, config.exchange says why and nothing is tried again. Last of all: creating and destroying an RCCL communicator leaves the process with slower host-side event waits -- measured, the
        # synchronous one-frame call went from 0.149 to 0.649 ms when this ran ahead of the extra rows, profiles/r06k_bench.json)
        line["config"]["exchange_selfcheck"] = exchange_selfcheck()
    if rank == 0 and args.probe_run:
        # a probe line carries timings only: nothing in it can be mistaken for the metric
        print(json.dumps({"probe_run": True, "debug_skip": os.environ.get("CIMBAR_HIP_DEBUG_SKIP", "0"), "payload_ok": ok,
                          "ms_per_step": line["ms_per_step"], "stage_ms": line["stage_ms"], "no_pipeline": line.get("no_pipeline")}), flush=True)
    elif rank == 0:
        line["summary"] = summary(line)          # last key: the driver keeps the tail of the line
        print(json.dumps(line), flush=True)
    if exchange is not None:
        exchange.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
