"""bench.py --config 4: BASELINE configs[3] -- an 8192-frame fountain stream of a 16 MiB file, the frames SPLIT over the ranks (strong
scaling) in 1024-frame slabs, every slab's chunks gathered to rank 0 over RCCL, and rank 0's single wirehair sink fed INSIDE the timed
region; the recovered file's SHA-256 is checked against the input.

The sink is the reference's own fountain_decoder_sink (out of this framework's scope by SURVEY 8: wirehair is a sequential sparse
solve that stays on the rank-0 host) -- here it is the consumer the decoder feeds, taken from the reference build oracle/_ref where that
exists; without it the run reports decode + gather only and checks the gathered chunks against the stream that was encoded.
"""
import ctypes
import hashlib
import os
import time

import numpy as np
import torch

from libcimbar_amd import framegen, modeb, multigpu

FILE_SIZE = 16 << 20
N_FRAMES = 8192
SLAB = 1024


def _ref():
    try:
        from oracle import pyref          # the reference build: input manufacture + the sink on rank 0 (consumer, not the product)
        return pyref.ref_lib(), pyref.P
    except Exception:
        return None, None


def bench(dec, dev, rank, world, args):
    import torch.distributed as dist
    n_frames = N_FRAMES if args.frames == 1024 else args.frames * 8       # --frames scales the job down for smoke runs (frames per slab)
    slab = min(SLAB, max(1, n_frames // 8))
    data = np.random.default_rng(4321).integers(0, 256, FILE_SIZE if n_frames == N_FRAMES else max(1 << 16, n_frames * 3000), dtype=np.uint8)
    ref, P = _ref()
    lo, hi, per = multigpu.shard_range(n_frames, rank, world)
    mine = hi - lo
    # the fountain stream: real wirehair chunks where the reference build is at hand, else a synthetic header + random payload
    if ref is not None:
        chunks_in = np.zeros((n_frames * 12, 625), np.uint8)
        assert ref.ref_fountain_chunks(P(data), data.size, 9, n_frames * 12, P(chunks_in)) == n_frames * 12
        payload_all = torch.from_numpy(chunks_in.reshape(n_frames, 7500))
    else:
        payload_all = framegen.synth_payload(n_frames, seed=4321, encode_id=9, file_size=int(data.size))
    payload = payload_all[lo:hi].to(dev)
    frames = torch.empty((mine, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    for a in range(0, mine, slab):
        b = min(mine, a + slab)
        dec.encode_batch_device(payload[a:b].contiguous().data_ptr(), b - a, frames[a:b].data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)

    chunks = torch.zeros((per, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros((per,), dtype=torch.int32, device=dev)
    gathered = (torch.zeros((world * per, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev),
                torch.zeros((world * per,), dtype=torch.int32, device=dev)) if rank == 0 else None
    out = np.zeros(data.size, np.uint8)

    def one_pass(timed):
        dec.reset_ccm()
        masks.zero_()
        t = {}
        t0 = time.perf_counter()
        for a in range(0, mine, slab):                       # consecutive slabs through the pipelined entry point
            b = min(mine, a + slab)
            dec.decode_batch_pipelined(frames[a:b].data_ptr(), b - a, chunks[a:b].data_ptr(), masks[a:b].data_ptr(), False, 2, stream.cuda_stream)
        dec.pipeline_wait(stream.cuda_stream, 0)
        torch.cuda.synchronize(dev)
        t["decode"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        all_c, all_m = multigpu.gather_chunks(chunks, masks, dst=0, out=gathered) if world > 1 else (chunks, masks)
        torch.cuda.synchronize(dev)
        t["gather"] = time.perf_counter() - t0
        res = {"fed": 0, "file_id": 0}
        t0 = time.perf_counter()
        if rank == 0:
            hc = all_c.cpu().numpy()
            hm = all_m.cpu().numpy().astype(np.uint32)
            t["d2h"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            if ref is not None:
                # frame order: rank r's slab is frames [r * per, r * per + (its count))
                ref.ref_sink_reset(625)
                fid = ctypes.c_uint32(0)
                ref.ref_sink_feed_batch.restype = ctypes.c_int64
                res["fed"] = int(ref.ref_sink_feed_batch(P(hc), P(hm), hc.shape[0], 12, 625, P(out), out.size, ctypes.byref(fid)))
                res["file_id"] = int(fid.value)
            t["sink"] = time.perf_counter() - t0
            res["chunks"], res["masks"] = hc, hm
        return t, res

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(1, min(args.warmup, 2))):
        one_pass(False)
    barrier()
    steps = max(1, min(args.steps, 5))
    t0 = time.perf_counter()
    acc = {}
    for _ in range(steps):
        t, res = one_pass(True)
        for k, v in t.items():
            acc[k] = acc.get(k, 0.0) + v / steps
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return None
    ok_chunks = True
    for r in range(world):
        a, b, _ = multigpu.shard_range(n_frames, r, world)
        ok_chunks = ok_chunks and bool((res["chunks"][r * per:r * per + (b - a)] == payload_all[a:b].numpy()).all()) and \
            bool((res["masks"][r * per:r * per + (b - a)] == 0xFFF).all())
    sha_ok = None
    if ref is not None:
        sha_ok = res["file_id"] != 0 and hashlib.sha256(out.tobytes()).hexdigest() == hashlib.sha256(data.tobytes()).hexdigest()
    if not ok_chunks or sha_ok is False:
        raise SystemExit("bench --config 4: gathered chunks / recovered file differ from the input")
    per_step = elapsed / steps
    return {
        "metric": "decoded cimbar frames/s (1024x1024 mode-B)", "value": round(n_frames / per_step, 1), "unit": "frames/s", "n_gpus": world,
        "steps": steps, "warmup": max(1, min(args.warmup, 2)), "ms_per_step": round(per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {n_frames}-frame fountain stream of a {data.size}-byte file split over {world} rank(s) in "
                               f"{slab}-frame slabs, gather to rank 0, single wirehair sink fed inside the timed region",
                   "frames_total": n_frames, "frames_per_rank": per, "parallelism": f"frame-sharded x{world}, RCCL gather to rank 0"},
        "stage_s": {k: round(v, 5) for k, v in acc.items()},
        "decode_gather_frames_per_s": round(n_frames / (acc["decode"] + acc["gather"]), 1),
        "sink": {"kind": "reference fountain_decoder_sink (oracle/_ref)" if ref is not None else None, "chunks_fed": res["fed"],
                 "chunks_per_s": round(res["fed"] / acc["sink"], 1) if ref is not None and acc.get("sink") else None,
                 "file_recovered_sha256_match": sha_ok},
        "chunks_match_encoded_stream": ok_chunks,
    }
