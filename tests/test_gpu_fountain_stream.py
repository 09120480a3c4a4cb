"""GPU, BASELINE configs[3] in miniature on one device: a real wirehair fountain stream of a file is rendered into frames,
decoded in slabs (as the ranks of an 8-GPU job would), the chunk slots are concatenated in rank order and fed to ONE reference
fountain_decoder_sink, which must reassemble the file."""
import ctypes
import hashlib

import numpy as np
import pytest
import torch

from libcimbar_amd import framegen, modeb, multigpu
from oracle import pyref
from oracle.pyref import P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,n_frames", [(1 << 20, 160), (16 << 20, 8192)], ids=["1MiB-160frames", "config4-16MiB-8192frames"])
def test_fountain_stream_through_sharded_decode_and_single_sink(hip_decoder, ref, size, n_frames):
    """the second case is BASELINE configs[3] at full size (16 MiB file, 8192 frames = 98 304 chunks whose 16-bit block ids wrap after
    frame 5461, eight slabs of 1024 frames), the eight ranks played one after the other by the one GPU of the test box"""
    dev = torch.device("cuda", 0)
    data = np.random.default_rng(4321).integers(0, 256, size, dtype=np.uint8)       # 1 MiB file -> 1695 wirehair blocks
    world = 8
    chunks_in = np.zeros((n_frames * 12, 625), np.uint8)
    assert ref.ref_fountain_chunks(P(data), data.size, 9, n_frames * 12, P(chunks_in)) == n_frames * 12
    payload = torch.from_numpy(chunks_in.reshape(n_frames, 7500)).to(dev)
    outs, masks_all = [], []
    for rank in range(world):                                   # what rank `rank` of an 8-GPU job would do
        lo, hi, per = multigpu.shard_range(n_frames, rank, world)
        frames = torch.empty((hi - lo, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
        hip_decoder.encode_batch_device(payload[lo:hi].contiguous().data_ptr(), hi - lo, frames.data_ptr(), torch.cuda.current_stream().cuda_stream)
        c = torch.zeros((per, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
        m = torch.zeros((per,), dtype=torch.int32, device=dev)
        dec = hip_decoder
        dec.reset_ccm()                                         # every rank starts with a fresh decoder
        dec.decode_batch_device(frames.data_ptr(), hi - lo, c.data_ptr(), m.data_ptr(), False, 2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(c)
        masks_all.append(m)
        del frames
    all_chunks, all_masks = torch.cat(outs, 0), torch.cat(masks_all, 0)     # == gather to rank 0 in rank order
    assert bool((all_chunks[:n_frames] == payload).all())

    ref.ref_sink_reset(625)
    out = np.zeros(data.size, np.uint8)
    done = []

    def on_complete(file_id):
        assert ref.ref_sink_recover(ctypes.c_uint32(file_id), P(out), out.size) == 1
        done.append(file_id)
    res = multigpu.feed_sink(lambda c: ref.ref_sink_decode_frame(P(np.ascontiguousarray(c)), 625), all_chunks, all_masks, on_complete)
    assert len(done) == 1, res[:10]
    assert hashlib.sha256(out.tobytes()).hexdigest() == hashlib.sha256(data.tobytes()).hexdigest()
    assert sum(1 for r in res if r == -1) > 0          # chunks after completion are ignored (fountain_decoder_sink.h:146-148)


def test_library_gather_with_one_rank(hip_decoder):
    """cimbar_hip_gather_chunks (ncclGather over RCCL, bound by the library itself) with a one-rank communicator: the gathered buffers are
    the rank's own chunks and masks -- the N = 1 case of the exchange a C++ host runs per batch (the N > 1 path needs an N-GPU node)"""
    from libcimbar_amd import decoder as D
    dev = torch.device("cuda", 0)
    n = 5
    g = torch.Generator(device="cpu").manual_seed(3)
    chunks = torch.randint(0, 256, (n, modeb.FRAME_BYTES), dtype=torch.uint8, generator=g).to(dev)
    masks = torch.tensor([0xFFF, 0x8e0, 0, 0x3cf, 1], dtype=torch.int32, device=dev)
    all_c = torch.zeros_like(chunks)
    all_m = torch.zeros_like(masks)
    uid = D.comm_unique_id()
    comm = hip_decoder.comm_init_rank(uid, 1, 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    hip_decoder.gather_chunks(comm, 0, chunks.data_ptr(), masks.data_ptr(), n, all_c.data_ptr(), all_m.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool((all_c == chunks).all()) and bool((all_m == masks).all())
    D.comm_destroy(comm)


def test_bench_config4_small(hip_decoder, ref):
    """bench.py --config 4 at a reduced size on one GPU: slabs through the pipelined entry point, sink fed, file recovered"""
    import argparse
    import bench_config4 as config4
    dev = torch.device("cuda", 0)
    line = config4.bench(hip_decoder, dev, 0, 1, argparse.Namespace(frames=32, steps=1, warmup=1))
    assert line["chunks_match_encoded_stream"] and line["sink"]["file_recovered_sha256_match"] is True
    assert line["scaling"] == "strong" and line["config"]["frames_total"] == 256
