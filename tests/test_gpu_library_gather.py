"""GPU: the exchange bench.py runs for N > 1 (libcimbar_amd.multigpu.LibraryGather -> cimbar_hip_gather_chunks, RCCL issued by the library on
a side stream, torch.distributed as the rendezvous) driven with a world of ONE rank -- every call of the N-rank path except the peers: the
communicator id through broadcast_object_list, ncclCommInitRank, the gather into the destination buffers, the event hand-over back to the
caller's stream, close()."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_library_gather_with_one_rank(hip_decoder):
    import torch.distributed as dist
    from libcimbar_amd import modeb, multigpu
    dev = torch.device("cuda", 0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        ex = multigpu.LibraryGather(hip_decoder, dev)
        g = torch.Generator(device="cpu").manual_seed(5)
        n = 37
        for rep in range(3):
            chunks = torch.randint(0, 256, (n, modeb.FRAME_BYTES), dtype=torch.uint8, generator=g).to(dev)
            masks = torch.randint(0, 4096, (n,), dtype=torch.int32, generator=g).to(dev)
            out = (torch.zeros_like(chunks), torch.zeros_like(masks)) if rep else None     # with and without preallocated destinations
            all_c, all_m, works = ex(chunks, masks, dst=0, out=out, async_op=True)
            for w in works:
                w.wait()
            torch.cuda.synchronize(dev)
            assert (all_c == chunks).all() and (all_m == masks).all()
        # the synchronous form
        all_c, all_m = ex(chunks, masks, dst=0, async_op=False)
        torch.cuda.synchronize(dev)
        assert (all_c == chunks).all() and (all_m == masks).all()
        ex.close()
        ex.close()          # (idempotent)
    finally:
        if created:
            dist.destroy_process_group()


def test_pipeline_gather_rides_on_the_batch_stream(hip_decoder, synth):
    """cimbar_hip_pipeline_gather (round 6): the exchange of a pipelined batch enqueued behind it on the batch's OWN stream -- what bench.py's N > 1
    loop issues per step -- with a one-rank communicator, through multigpu.StepPipeline's inline mode: several steps in flight, every step's gathered
    chunks and masks equal to what the same frames decode to through the ordinary call, after nothing but pipeline_wait."""
    from libcimbar_amd import modeb, multigpu
    from tests import frames as F
    dev = torch.device("cuda", 0)
    payload, fr = F.clean_frames(synth, 6, seed=321)
    batches = [np.ascontiguousarray(np.stack([fr[(k + j) % 6] if (k + j) % 4 else F.add_noise(fr[(k + j) % 6], 30, k + j) for j in range(5)])) for k in range(7)]
    want = []
    hip_decoder.reset_ccm()
    for b in batches:
        total, chunks, masks = hip_decoder.decode_batch(b)
        want.append((chunks.reshape(len(b), -1).copy(), masks.copy()))
    dbatches = [torch.from_numpy(b).to(dev) for b in batches]
    D = hip_decoder.pipeline_depth
    n = 5
    outs = [(torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros((n,), dtype=torch.int32, device=dev)) for _ in range(D)]
    gathered = [(torch.zeros_like(c), torch.zeros_like(m)) for c, m in outs]
    stream = torch.cuda.current_stream(dev)
    ex = multigpu.LibraryGather(hip_decoder, dev)          # (no process group: a job of one process, one rank)
    assert ex.nranks == 1

    def issue(b, k):
        hip_decoder.decode_batch_pipelined(dbatches[k].data_ptr(), n, outs[b][0].data_ptr(), outs[b][1].data_ptr(), False, 2, stream.cuda_stream)

    def ready(keep_newest):
        hip_decoder.pipeline_wait(stream.cuda_stream, keep_newest=keep_newest)

    hip_decoder.reset_ccm()
    pipe = multigpu.StepPipeline(outs, D, issue, ready, gathered=gathered, dst=0, gather=ex, inline_gather=ex.inline)
    got = []
    for k in range(len(batches)):
        pipe.step()
        if k >= D - 1:                                       # step k-D+1 is complete, gathered chunks included, once the stream has passed this wait
            ready(D - 1)
            j = k - (D - 1)
            got.append((gathered[j % D][0].clone(), gathered[j % D][1].clone()))
    pipe.drain()
    for j in range(len(batches) - (D - 1), len(batches)):
        got.append((gathered[j % D][0].clone(), gathered[j % D][1].clone()))
    torch.cuda.synchronize(dev)
    assert pipe.gathers == len(batches) and len(got) == len(batches)
    for k, (c, m) in enumerate(got):
        assert (c.cpu().numpy() == want[k][0]).all() and (m.cpu().numpy().astype(np.uint32) == want[k][1]).all(), k
    ex.close()
    # without a pipelined batch on the context the call is refused
    from libcimbar_amd import HipDecoder, decoder as D_
    fresh = HipDecoder(0)
    comm = fresh.comm_init_rank(D_.comm_unique_id(), 1, 0)
    with pytest.raises(D_.CimbarHipError):
        fresh.pipeline_gather(comm, 0, outs[0][0].data_ptr(), outs[0][1].data_ptr(), n, gathered[0][0].data_ptr(), gathered[0][1].data_ptr())
    D_.comm_destroy(comm)
    fresh.close()
