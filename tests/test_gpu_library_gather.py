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
