"""The N>1 path on CPU: world_size-2 `gloo`, frame-sharded decode + gather of the chunk slots to rank 0 + single sink.
The per-rank decode is the oracle here (no GPU in this container); on the GPU box bench.py runs the same
libcimbar_amd.multigpu code with the HIP decoder and the `nccl` (RCCL) backend."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libcimbar_amd import framegen, modeb, multigpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, use_fountain, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyref
        synth = framegen.FrameSynth("cpu")
        data = np.random.default_rng(4321).integers(0, 256, 20000, dtype=np.uint8)
        if use_fountain:
            L = pyref.ref_lib()
            chunks = np.zeros((n_frames * 12, 625), np.uint8)
            assert L.ref_fountain_chunks(pyref.P(data), data.size, 5, n_frames * 12, pyref.P(chunks)) == n_frames * 12
            payload = torch.from_numpy(chunks.reshape(n_frames, 7500))
        else:
            payload = framegen.synth_payload(n_frames, seed=99)
        lo, hi, per = multigpu.shard_range(n_frames, rank, world)
        frames = synth.frames_from_payload(payload[lo:hi]).numpy()
        out_chunks = torch.zeros((per, modeb.FRAME_BYTES), dtype=torch.uint8)
        out_masks = torch.zeros((per,), dtype=torch.int32)
        ccm = pyref.CoCcm()
        for k in range(hi - lo):   # stand-in for HipDecoder.decode_batch_device on this rank's GPU
            r, c, m, ccm = pyref.oracle_decode(frames[k], 0, 2, ccm)
            out_chunks[k] = torch.from_numpy(c.reshape(-1))
            out_masks[k] = m
        all_chunks, all_masks = multigpu.gather_chunks(out_chunks, out_masks, dst=0)
        if rank == 0:
            assert all_chunks.shape == (world * per, modeb.FRAME_BYTES)
            assert (all_chunks[:n_frames] == payload).all()
            assert (all_masks[:n_frames] == 0xFFF).all() and (all_masks[n_frames:] == 0).all()
            order = []
            if use_fountain:
                L.ref_sink_reset(625)
                L.ref_sink_decode_frame.restype = ctypes.c_int64
                out = np.zeros(20000, np.uint8)
                recovered = []

                def on_complete(file_id):
                    assert L.ref_sink_recover(ctypes.c_uint32(file_id), pyref.P(out), 20000) == 1
                    recovered.append(file_id)
                res = multigpu.feed_sink(lambda c: L.ref_sink_decode_frame(pyref.P(np.ascontiguousarray(c)), 625), all_chunks, all_masks, on_complete)
                assert len(recovered) == 1, res
                assert (out == data).all()
                assert L.ref_sink_is_done(ctypes.c_uint32(recovered[0])) == 1
            else:
                multigpu.feed_sink(lambda c: order.append(bytes(c[4:6])) or 0, all_chunks, all_masks)
                ids = [int.from_bytes(b, "big") for b in order]
                assert ids == list(range(n_frames * 12)), "sink must see chunks in frame order, then chunk order"
            open(result_path, "w").write("ok")
        else:
            assert all_chunks is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [4, 3])
def test_sharded_decode_gathers_in_frame_order(tmp_path, n_frames):
    res = tmp_path / "ok"
    mp.spawn(_worker, args=(2, _free_port(), n_frames, False, str(res)), nprocs=2, join=True)
    assert res.read_text() == "ok"


def test_fountain_stream_reassembles_through_single_sink(tmp_path, ref):
    # BASELINE config 4 in miniature: a wirehair stream sharded over 2 ranks, gathered, decoded by ONE fountain_decoder_sink
    res = tmp_path / "ok"
    mp.spawn(_worker, args=(2, _free_port(), 4, True, str(res)), nprocs=2, join=True)
    assert res.read_text() == "ok"


def test_shard_range():
    assert [multigpu.shard_range(10, r, 4)[:2] for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert multigpu.shard_range(8192, 7, 8)[:2] == (7168, 8192)


def _pipeline_worker(rank, world, port, depth, nsteps, result_path):
    """bench.py's timed loop in miniature: `depth` steps in flight, each step's outputs gathered depth-1 steps later"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 3
        nbuf = max(depth, 2)
        outs = [(torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8), torch.zeros((n,), dtype=torch.int32)) for _ in range(nbuf)]
        gathered = [(torch.zeros((world * n, modeb.FRAME_BYTES), dtype=torch.uint8), torch.zeros((world * n,), dtype=torch.int32))
                    if rank == 0 else None for _ in range(nbuf)]
        issued, waits, seen = [], [], []

        def issue(b, k):           # stand-in for decode_batch_pipelined: the "decoded" bytes name the step and the rank
            outs[b][0].fill_((k * 7 + rank) % 251)
            outs[b][1].fill_(k * 100 + rank)
            issued.append((b, k))

        def ready(keep_newest):
            waits.append(keep_newest)

        pipe = multigpu.StepPipeline(outs, depth, issue, ready, gathered=gathered, dst=0)
        orig_gather = pipe._gather

        def spy(b):
            before = pipe.gathers
            orig_gather(b)
            if pipe.gathers > before and rank == 0:
                for w in pipe.pending[b]:
                    w.wait()
                seen.append((int(pipe.last[1][0]) // 100, pipe.last[0][:, 0].clone(), pipe.last[1].clone()))
        pipe._gather = spy
        for _ in range(nsteps):
            pipe.step()
        pipe.drain()
        assert pipe.gathers == nsteps, "every step is gathered exactly once"
        assert [k for _, k in issued] == list(range(nsteps))
        assert waits[-1] == 0 and all(w == depth - 1 for w in waits[:-1])
        if rank == 0:
            assert [s for s, _, _ in seen] == list(range(nsteps)), "steps reach rank 0 in order"
            for s, c0, m in seen:
                want_m = torch.tensor([s * 100 + r for r in range(world) for _ in range(n)], dtype=torch.int32)
                want_c = torch.tensor([(s * 7 + r) % 251 for r in range(world) for _ in range(n)], dtype=torch.uint8)
                assert (m == want_m).all() and (c0 == want_c).all(), s
            open(result_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth,nsteps", [(4, 11), (2, 5), (1, 3), (4, 2)])
def test_step_pipeline_gathers_every_step_once_and_in_order(tmp_path, depth, nsteps):
    res = tmp_path / "ok"
    mp.spawn(_pipeline_worker, args=(2, _free_port(), depth, nsteps, str(res)), nprocs=2, join=True)
    assert res.read_text() == "ok"


def _inline_worker(rank, world, port, depth, nsteps, result_path):
    """the same loop with the exchange riding on each step (StepPipeline's inline mode: what bench.py --gpus N issues since round 6 through
    cimbar_hip_pipeline_gather); the stand-in gathers synchronously over gloo, which is what "in the step's own stream order" means on a CPU"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 3
        outs = [(torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8), torch.zeros((n,), dtype=torch.int32)) for _ in range(depth)]
        gathered = [(torch.zeros((world * n, modeb.FRAME_BYTES), dtype=torch.uint8), torch.zeros((world * n,), dtype=torch.int32))
                    if rank == 0 else None for _ in range(depth)]
        issued, waits, calls = [], [], []

        def issue(b, k):
            outs[b][0].fill_((k * 7 + rank) % 251)
            outs[b][1].fill_(k * 100 + rank)
            issued.append((b, k))

        def ready(keep_newest):
            waits.append(keep_newest)

        def inline(chunks, masks, dst=0, out=None):
            calls.append(len(issued) - 1)                # issued right behind the step, before the next one
            return multigpu.gather_chunks(chunks, masks, dst=dst, out=out)

        pipe = multigpu.StepPipeline(outs, depth, issue, ready, gathered=gathered, dst=0, inline_gather=inline)
        seen = []
        for k in range(nsteps):
            pipe.step()
            if rank == 0:
                seen.append((pipe.last[0][:, 0].clone(), pipe.last[1].clone()))
        pipe.drain()
        assert pipe.gathers == nsteps and calls == list(range(nsteps)), "every step is exchanged exactly once, right behind its issue"
        assert waits == [0], "no wait is needed before an exchange that rides on its step: only the drain waits"
        assert [b for b, _ in issued] == [k % depth for k in range(nsteps)], "a buffer set always returns to the same place in the rotation (= the same pipeline stream)"
        if rank == 0:
            for s, (c0, m) in enumerate(seen):
                want_m = torch.tensor([s * 100 + r for r in range(world) for _ in range(n)], dtype=torch.int32)
                want_c = torch.tensor([(s * 7 + r) % 251 for r in range(world) for _ in range(n)], dtype=torch.uint8)
                assert (m == want_m).all() and (c0 == want_c).all(), s
            open(result_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth,nsteps", [(3, 10), (2, 5), (4, 3)])
def test_step_pipeline_inline_exchange_rides_on_every_step(tmp_path, depth, nsteps):
    res = tmp_path / "ok"
    mp.spawn(_inline_worker, args=(2, _free_port(), depth, nsteps, str(res)), nprocs=2, join=True)
    assert res.read_text() == "ok"


def test_inline_exchange_wants_one_buffer_set_per_step_in_flight():
    outs = [(torch.zeros((1, modeb.FRAME_BYTES), dtype=torch.uint8), torch.zeros((1,), dtype=torch.int32)) for _ in range(5)]
    with pytest.raises(AssertionError):
        multigpu.StepPipeline(outs, 3, lambda b, k: None, lambda keep: None, inline_gather=lambda *a, **k: (None, None))
