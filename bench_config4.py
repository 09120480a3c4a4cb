"""bench.py --config 4: BASELINE configs[3] -- an 8192-frame fountain stream of a 16 MiB file, the frames SPLIT over the ranks (strong
scaling) in 1024-frame slabs, every slab's chunks gathered to rank 0 over RCCL, and rank 0's single wirehair sink fed INSIDE the timed
region; the recovered file's SHA-256 is checked against the input.

The sink is the reference's own fountain_decoder_sink (out of this framework's scope by SURVEY 8: wirehair is a sequential sparse
solve that stays on the rank-0 host) -- here it is the consumer the decoder feeds, taken from the reference build oracle/_ref where that
exists; without it the run reports decode + gather only and checks the gathered chunks against the stream that was encoded.

Shape of one pass (round 3): nothing waits for anything it does not need.
  * every rank decodes its slabs back to back through the pipelined entry point;
  * slab s is gathered by the LIBRARY's exchange (cimbar_hip_gather_chunks: ncclGather issued by libcimbar_hip.so) on a side stream as soon
    as it is complete, while slab s+1 decodes;
  * on rank 0 the gathered slab goes to pinned host memory on a copy stream of its own and a host thread feeds it to the sink while the
    GPU is already on the next slabs -- the shape of the reference's receive loop, where decoder threads hand chunks to one sink through
    concurrent_fountain_decoder_sink (fountain/concurrent_fountain_decoder_sink.h:58-90);
  * once the sink reports the file complete the feeder stops handing chunks over (the reference's sink ignores them anyway: is_done,
    fountain_decoder_sink.h:146-148) -- the remaining slabs are still decoded and gathered, they are part of the job.
The timed region ends when every slab has been decoded and gathered AND the sink has the file.
"""
import ctypes
import hashlib
import threading
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


def bench(dec, dev, rank, world, args, quiet=False):
    import torch.distributed as dist
    n_frames = N_FRAMES if args.frames == 1024 else args.frames * 8       # --frames scales the job down for smoke runs (frames per slab)
    slab = min(SLAB, max(1, n_frames // 8))
    data = np.random.default_rng(4321).integers(0, 256, FILE_SIZE if n_frames == N_FRAMES else max(1 << 16, n_frames * 3000), dtype=np.uint8)
    ref, P = _ref()
    lo, hi, per = multigpu.shard_range(n_frames, rank, world)
    mine = hi - lo
    nslabs = (per + slab - 1) // slab                                   # the same on every rank (short last shards pad with undelivered frames)
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

    chunks = torch.zeros((nslabs * slab, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)     # frames beyond `mine` stay undelivered (mask 0)
    masks = torch.zeros((nslabs * slab,), dtype=torch.int32, device=dev)
    exchange, exchange_name = None, None
    if world > 1:
        try:
            exchange = multigpu.LibraryGather(dec, dev)
            exchange_name = "cimbar_hip_gather_chunks (RCCL ncclGather issued by the library)"
        except Exception as e:
            exchange_name = f"torch.distributed.gather (library exchange unavailable: {e!r})"
        flags = torch.tensor([1 if exchange is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if int(flags.item()) == 0 and exchange is not None:
            exchange.close()
            exchange, exchange_name = None, "torch.distributed.gather (library exchange unavailable on another rank)"
    # rank 0: one gathered buffer + one pinned host buffer per slab (61 MB each way for the whole job), a copy stream, the feeder thread
    if rank == 0:
        g_chunks = [torch.zeros((world * slab, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev) for _ in range(nslabs)] if world > 1 else None
        g_masks = [torch.zeros((world * slab,), dtype=torch.int32, device=dev) for _ in range(nslabs)] if world > 1 else None
        h_chunks = [torch.zeros((world * slab, modeb.FRAME_BYTES), dtype=torch.uint8).pin_memory() for _ in range(nslabs)]
        h_masks = [torch.zeros((world * slab,), dtype=torch.int32).pin_memory() for _ in range(nslabs)]
        copy_stream = torch.cuda.Stream(dev)
    out = np.zeros(data.size, np.uint8)

    def one_pass():
        dec.reset_ccm()
        masks.zero_()
        torch.cuda.synchronize(dev)
        t = {}
        res = {"fed": 0, "file_id": 0, "slabs_fed": 0, "sink_s": 0.0, "feed_s": 0.0, "solve_s": 0.0, "done_at_s": None}
        copied = [torch.cuda.Event() for _ in range(nslabs)] if rank == 0 else None
        issued = threading.Semaphore(0)

        def feeder(t_start):
            # the consumer side of the reference's worker pool -> one sink (web/recv-worker.js:47-64; concurrent_fountain_decoder_sink.h:58-90)
            if ref is not None:
                ref.ref_sink_reset(625)
                ref.ref_sink_feed_batch_timed.restype = ctypes.c_int64
            fid = ctypes.c_uint32(0)
            feed_s, solve_s = ctypes.c_double(0.0), ctypes.c_double(0.0)
            for s in range(nslabs):
                issued.acquire()
                if res["file_id"]:
                    continue                                   # complete: the sink would ignore these chunks (is_done)
                copied[s].synchronize()
                if ref is None:
                    continue
                t0 = time.perf_counter()
                hc, hm = h_chunks[s].numpy(), h_masks[s].numpy().view(np.uint32)
                res["fed"] += int(ref.ref_sink_feed_batch_timed(P(hc), P(hm), hc.shape[0], 12, 625, P(out), out.size, ctypes.byref(fid),
                                                               ctypes.byref(feed_s), ctypes.byref(solve_s)))
                res["sink_s"] += time.perf_counter() - t0
                res["feed_s"], res["solve_s"] = feed_s.value, solve_s.value
                res["slabs_fed"] += 1
                if fid.value:
                    res["file_id"] = int(fid.value)
                    res["done_at_s"] = time.perf_counter() - t_start

        t_start = time.perf_counter()
        th = None
        if rank == 0:
            th = threading.Thread(target=feeder, args=(t_start,))
            th.start()
        pend = []
        for s in range(nslabs):                                # consecutive slabs through the pipelined entry point
            a = s * slab
            n_here = max(0, min(mine - a, slab))
            if n_here > 0:
                dec.decode_batch_pipelined(frames[a:a + n_here].data_ptr(), n_here, chunks[a:a + n_here].data_ptr(), masks[a:a + n_here].data_ptr(), False, 2, stream.cuda_stream)
            # slab s-1 is complete once the pipeline holds only the newest batch: gather it and ship it while slab s decodes
            for q in ([s - 1] if s > 0 else []) + ([s] if s == nslabs - 1 else []):
                dec.pipeline_wait(stream.cuda_stream, 0 if q == s else 1)
                c, m = chunks[q * slab:(q + 1) * slab], masks[q * slab:(q + 1) * slab]
                if world > 1:
                    if exchange is not None:
                        _c, _m, works = exchange(c, m, dst=0, out=(g_chunks[q], g_masks[q]) if rank == 0 else None, async_op=True)
                    else:
                        _c, _m, works = multigpu.gather_chunks(c, m, dst=0, out=(g_chunks[q], g_masks[q]) if rank == 0 else None, async_op=True)
                    pend.append(works)
                    if rank == 0:
                        for w in works:
                            if w is not None:
                                with torch.cuda.stream(copy_stream):
                                    w.wait()
                        c, m = g_chunks[q], g_masks[q]
                if rank == 0:
                    ready = torch.cuda.Event()
                    ready.record(stream)
                    copy_stream.wait_event(ready)
                    with torch.cuda.stream(copy_stream):
                        h_chunks[q].copy_(c, non_blocking=True)
                        h_masks[q].copy_(m, non_blocking=True)
                        copied[q].record(copy_stream)
                    issued.release()
        for works in pend:
            for w in works:
                if w is not None:
                    w.wait()
        torch.cuda.synchronize(dev)
        t["decode_gather_d2h"] = time.perf_counter() - t_start
        if th is not None:
            th.join()
        t["total"] = time.perf_counter() - t_start
        t["sink_busy"] = res["sink_s"]
        t["sink_feed"] = res["feed_s"]
        t["sink_solve"] = res["solve_s"]
        return t, res

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def decode_only():
        """the same slabs without exchange, copy or sink: what the GPUs alone sustain on this job"""
        dec.reset_ccm()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for s in range(nslabs):
            a = s * slab
            n_here = max(0, min(mine - a, slab))
            if n_here > 0:
                dec.decode_batch_pipelined(frames[a:a + n_here].data_ptr(), n_here, chunks[a:a + n_here].data_ptr(), masks[a:a + n_here].data_ptr(), False, 2, stream.cuda_stream)
        dec.pipeline_wait(stream.cuda_stream, 0)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    for _ in range(max(1, min(args.warmup, 2))):
        one_pass()
    barrier()
    steps = max(1, min(args.steps, 5))
    t0 = time.perf_counter()
    acc = {}
    for _ in range(steps):
        t, res = one_pass()
        for k, v in t.items():
            acc[k] = acc.get(k, 0.0) + v / steps
        if world > 1:
            dist.barrier()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    d_only = min(decode_only() for _ in range(3))
    if world > 1:
        tt = torch.tensor([d_only], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        d_only = float(tt.item())
    if exchange is not None:
        exchange.close()
    if rank != 0:
        return None
    # what rank 0 received: slab q of rank r = frames [r * per + q * slab, ...)
    ok_chunks = True
    for q in range(nslabs):
        hc, hm = h_chunks[q].numpy(), h_masks[q].numpy().view(np.uint32)
        for r in range(world):
            a, b, _ = multigpu.shard_range(n_frames, r, world)
            fa = a + q * slab
            cnt = max(0, min(b - fa, slab))
            if cnt:
                ok_chunks = ok_chunks and bool((hc[r * slab:r * slab + cnt] == payload_all[fa:fa + cnt].numpy()).all()) and bool((hm[r * slab:r * slab + cnt] == 0xFFF).all())
    sha_ok = None
    if ref is not None:
        sha_ok = res["file_id"] != 0 and hashlib.sha256(out.tobytes()).hexdigest() == hashlib.sha256(data.tobytes()).hexdigest()
    if not ok_chunks or sha_ok is False:
        raise SystemExit("bench --config 4: gathered chunks / recovered file differ from the input")
    per_step = elapsed / steps
    return {
        "metric": "decoded cimbar frames/s (1024x1024 mode-B)", "value": round(n_frames / per_step, 1),
        # `value` includes wirehair's one sequential solve on a rank-0 host thread (the reference's sink, the same at every N); this is the step without it
        "value_without_solve": round(n_frames / max(per_step - acc.get("sink_solve", 0.0), 1e-9), 1), "unit": "frames/s", "n_gpus": world,
        "steps": steps, "warmup": max(1, min(args.warmup, 2)), "ms_per_step": round(per_step * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {n_frames}-frame fountain stream of a {data.size}-byte file split over {world} rank(s) in "
                               f"{slab}-frame slabs, gather to rank 0, single wirehair sink fed inside the timed region (slab s feeds the sink "
                               "while slab s+1 decodes; feeding stops when the file is complete). `value` is bounded by the reference sink's sequential wirehair solve "
                               "on one host thread; `value_without_solve` subtracts the sink's mean solve time in full -- an UPPER bound for the part that scales with GPUs "
                               "(the sink is fed while the next slab decodes, so some of that time is already overlapped)",
                   "frames_total": n_frames, "frames_per_rank": per, "exchange": exchange_name,
                   "parallelism": f"frame-sharded x{world}, RCCL gather to rank 0"},
        "stage_s": {k: round(v, 5) for k, v in acc.items()},
        "decode_only_frames_per_s": round(n_frames / d_only, 1),
        "end_to_end_over_decode_only": round((n_frames / per_step) / (n_frames / d_only), 3),
        # the same with wirehair's solve (one call of the reference's sink, on one host thread, the same at every N) taken out of the step: what the
        # decode + gather + copy + block-by-block feeding sustain -- the part of this row that can scale with the GPUs
        "without_solve": {"ms_per_step": round((per_step - acc.get("sink_solve", 0.0)) * 1e3, 3),
                          "frames_per_s": round(n_frames / max(per_step - acc.get("sink_solve", 0.0), 1e-9), 1),
                          "over_decode_only": round(d_only / max(per_step - acc.get("sink_solve", 0.0), 1e-9), 3)},
        "sink": {"kind": "reference fountain_decoder_sink (oracle/_ref)" if ref is not None else None, "chunks_fed": res["fed"], "slabs_fed": res["slabs_fed"],
                 "busy_s": round(res["sink_s"], 5), "feed_s": round(res["feed_s"], 5), "solve_s": round(res["solve_s"], 5), "file_complete_after_s": None if res["done_at_s"] is None else round(res["done_at_s"], 5),
                 "chunks_per_s": round(res["fed"] / res["sink_s"], 1) if ref is not None and res["sink_s"] else None,
                 "file_recovered_sha256_match": sha_ok,
                 "note": "feed_s = the block-by-block calls (wirehair keeps its system up to date as blocks arrive), solve_s = the one call that completes "
                         "the 27 104-block file: a sequential solve on one host thread, what the job waits for once the GPUs are done, at every N; "
                         "feeding stops at that chunk"},
        "chunks_match_encoded_stream": ok_chunks,
    }
