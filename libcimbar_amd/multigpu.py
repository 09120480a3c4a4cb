"""Frame-parallel sharding across the GPUs of one node + gather of the decoded chunks to a single sink on rank 0.

Frames are independent units (reference TODO.md:15), so rank r decodes the contiguous slab [r*n/W, (r+1)*n/W) with no
data-path collective; the only exchange is one fixed-size gather per batch (n_r * (12*625 + 4) bytes per rank), which is
what the reference's own worker pool does with postMessage (web/recv-worker.js:47-64 -> web/recv.js:36). With the `nccl`
backend that gather is RCCL over xGMI; the CPU tests run the same code on `gloo`.
"""
import torch
import torch.distributed as dist

from . import modeb


def shard_range(n_frames, rank, world):
    """Contiguous slab of frames owned by `rank` (the last ranks get the shorter slabs when n % world != 0)."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    hi = min(lo + per, n_frames)
    return lo, hi, per


def gather_chunks(chunks, masks, dst=0, group=None, out=None, async_op=False):
    """chunks: (n_r, 7500) uint8, masks: (n_r,) int32 on every rank (same n_r everywhere). On `dst` returns
    (chunks (W*n_r, 7500), masks (W*n_r,)) in rank order == frame order; elsewhere (None, None).
    out: optional preallocated (chunks (W*n_r,7500), masks (W*n_r,)) on `dst` (no allocation / concatenation per call).
    async_op: return (all_chunks, all_masks, [work handles]) without waiting -- the caller overlaps the exchange with the next
    batch's decode and calls .wait() on the handles before touching the buffers again."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return (chunks, masks, []) if async_op else (chunks, masks)
    works = []
    if rank == dst:
        if out is None:
            n_r = chunks.shape[0]
            out = (torch.empty((world * n_r, chunks.shape[1]), dtype=chunks.dtype, device=chunks.device),
                   torch.empty((world * n_r,), dtype=masks.dtype, device=masks.device))
        all_c, all_m = out
        works.append(dist.gather(chunks, list(all_c.chunk(world, 0)), dst=dst, group=group, async_op=async_op))
        works.append(dist.gather(masks, list(all_m.chunk(world, 0)), dst=dst, group=group, async_op=async_op))
    else:
        all_c = all_m = None
        works.append(dist.gather(chunks, None, dst=dst, group=group, async_op=async_op))
        works.append(dist.gather(masks, None, dst=dst, group=group, async_op=async_op))
    return (all_c, all_m, works) if async_op else (all_c, all_m)


class _EventWork:
    """what StepPipeline needs of an exchange in flight: wait() orders the current stream behind it"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class LibraryGather:
    """The product's own exchange, cimbar_hip_gather_chunks (RCCL ncclGather issued by libcimbar_hip.so, csrc/comm.hip.inc), as a drop-in
    for gather_chunks(..., async_op=True): the gather runs on a side stream that waits for what the current stream has been told to wait
    for, so it overlaps the decode steps issued afterwards. torch.distributed only carries the 128-byte communicator id to the ranks
    (rendezvous); no torch collective touches the data path."""

    def __init__(self, dec, dev, group=None, copy_only=False):
        self.dec, self.dev, self.group = dec, dev, group
        self.copy_only = copy_only      # (experiment, one rank only: a plain device copy in ncclGather's place -- the same streams and events without RCCL's kernel)
        # (a job of one process needs no rendezvous: the communicator then has one rank and ncclGather is a copy on the device -- bench.py's N = 1 line
        # runs its steps through it so that the record the driver takes shows the library's RCCL path loaded and issued inside the timed loop)
        solo = not dist.is_initialized()
        self.world, self.rank = (1, 0) if solo else (dist.get_world_size(group), dist.get_rank(group))
        from . import decoder as _d
        box = [None]
        if self.rank == 0:
            try:
                box[0] = _d.comm_unique_id()
            except Exception as e:          # (no RCCL for the library to bind: every rank must learn that, or the others wait in the broadcast forever)
                box[0] = e
        if not solo:
            dist.broadcast_object_list(box, src=0, group=group)
        if isinstance(box[0], Exception):
            raise RuntimeError(f"rank 0 could not create a communicator id: {box[0]!r}")
        self.comm = dec.comm_init_rank(box[0], self.world, self.rank)
        # what RCCL says about the communicator it built (ncclCommCount / ncclCommUserRank): the bench line quotes THIS, not the launcher's WORLD_SIZE
        self.nranks, self.comm_rank = _d.comm_info(self.comm)
        if (self.nranks, self.comm_rank) != (self.world, self.rank):
            raise RuntimeError(f"RCCL communicator has {self.nranks} ranks (this one {self.comm_rank}), the job has {self.world} (this one {self.rank})")
        self.stream = torch.cuda.Stream(dev)

    def __call__(self, chunks, masks, dst=0, group=None, out=None, async_op=True):
        if self.rank == dst and out is None:
            n_r = chunks.shape[0]
            out = (torch.empty((self.world * n_r, chunks.shape[1]), dtype=chunks.dtype, device=chunks.device),
                   torch.empty((self.world * n_r,), dtype=masks.dtype, device=masks.device))
        all_c, all_m = out if self.rank == dst else (None, None)
        before = torch.cuda.Event()
        before.record(torch.cuda.current_stream(self.dev))
        self.stream.wait_event(before)
        if self.copy_only and self.world == 1:
            with torch.cuda.stream(self.stream):
                all_c.copy_(chunks, non_blocking=True)
                all_m.copy_(masks, non_blocking=True)
        else:
            self.dec.gather_chunks(self.comm, dst, chunks.data_ptr(), masks.data_ptr(), chunks.shape[0],
                                   all_c.data_ptr() if all_c is not None else 0, all_m.data_ptr() if all_m is not None else 0, self.stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        work = _EventWork(done)
        if not async_op:
            work.wait()
            return all_c, all_m
        return all_c, all_m, [work]

    def inline(self, chunks, masks, dst=0, out=None):
        """cimbar_hip_pipeline_gather: the exchange of the pipelined batch issued LAST (whose outputs `chunks` / `masks` are), enqueued on that batch's own
        stream -- no side stream, no event: HipDecoder.pipeline_wait covers the gathered buffers like the batch's outputs. Returns (all_chunks, all_masks)."""
        all_c, all_m = out if (self.rank == dst and out is not None) else (None, None)
        if self.rank == dst and out is None:
            raise ValueError("inline exchange: the root passes its receive buffers (one set per output buffer set)")
        self.dec.pipeline_gather(self.comm, dst, chunks.data_ptr(), masks.data_ptr(), chunks.shape[0],
                                 all_c.data_ptr() if all_c is not None else 0, all_m.data_ptr() if all_m is not None else 0)
        return all_c, all_m

    def close(self):
        from . import decoder as _d
        if self.comm is not None:
            torch.cuda.synchronize(self.dev)
            _d.comm_destroy(self.comm)
            self.comm = None


class StepPipeline:
    """Host-side bookkeeping for a continuous stream of decode steps on one rank (what bench.py's timed loop is):

      * `depth` steps may be in flight (HipDecoder.decode_batch_pipelined); step k writes output buffer set k % nbuf;
      * with world > 1 the outputs of step k-(depth-1) are gathered to rank 0 right after step k has been issued (RCCL runs the
        exchange on its own stream), so neither the decode pipeline nor the exchange ever waits for the other;
      * a buffer set is not handed to a new step before the exchange that last read it is over.

    issue(buf, step) enqueues one decode into outs[buf]; ready(keep_newest) makes the current stream wait for every issued step
    except the `keep_newest` most recent ones. Both are callables so that the CPU test can drive the same logic with gloo."""

    def __init__(self, outs, depth, issue, ready, gathered=None, dst=0, group=None, gather=None, ready_for_gather=None, inline_gather=None):
        self.outs, self.depth, self.issue, self.ready = outs, max(1, int(depth)), issue, ready
        # ready_for_gather(keep_newest): orders the EXCHANGE's own stream behind the issued steps instead of the caller's. Round 6, measured at N = 1 with
        # a one-rank communicator (profiles/r06b_bench_exchange_*): making the caller's stream wait for step k-depth+1 before every gather -- the stream
        # the next step's "frames are ready" event is recorded on -- cost the pipelined loop 11 % (0.759 -> 0.840 ms per step), with ncclGather or a plain
        # device copy in its place alike: the wait, not RCCL's kernel. Falls back to `ready` (the CPU test's callables, torch.distributed's gather).
        self.ready_for_gather = ready_for_gather
        # inline_gather(chunks, masks, dst=, out=): the exchange of the step just issued, enqueued behind it on the step's own stream (LibraryGather.inline
        # -> cimbar_hip_pipeline_gather). Nothing to wait for before or after: stream order does it all, provided a buffer set always returns to the same
        # pipeline stream (one set per step in flight), which is asserted here.
        self.inline_gather = inline_gather
        if inline_gather is not None:
            assert len(outs) == self.depth, "inline exchange: exactly one output buffer set per step in flight (a set then always rides the same stream)"
        self.gather_fn = gather if gather is not None else gather_chunks     # e.g. a LibraryGather: the library's own RCCL exchange
        self.nbuf = len(outs)
        assert self.nbuf >= self.depth, "one output buffer set per step in flight"
        self.gathered = gathered if gathered is not None else [None] * self.nbuf
        self.dst, self.group = dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchanging = self.world > 1 or gather is not None     # (an explicit exchange is issued at any job size, one rank included)
        self.pending = [[] for _ in range(self.nbuf)]
        self.fresh = [False] * self.nbuf
        self.steps = 0
        self.last = None               # (all_chunks, all_masks) of the most recent gather (rank `dst` only)
        self.gathers = 0

    def _gather(self, b):
        if not self.fresh[b]:
            return
        self.fresh[b] = False
        chunks, masks = self.outs[b]
        all_c, all_m, self.pending[b] = self.gather_fn(chunks, masks, dst=self.dst, group=self.group, out=self.gathered[b], async_op=True)
        self.last = (all_c, all_m)
        self.gathers += 1

    def step(self):
        k = self.steps
        b = k % self.nbuf
        self.steps += 1
        for w in self.pending[b]:      # the exchange that last used this buffer set must be over before it is overwritten
            if w is not None:
                w.wait()
        self.pending[b] = []
        self.issue(b, k)
        self.fresh[b] = True
        if self.inline_gather is not None:
            chunks, masks = self.outs[b]
            self.last = self.inline_gather(chunks, masks, dst=self.dst, out=self.gathered[b])
            self.fresh[b] = False
            self.gathers += 1
        elif self.exchanging and k >= self.depth - 1:
            (self.ready_for_gather or self.ready)(self.depth - 1)     # step k-depth+1 is complete from here on in (the exchange's) stream order
            self._gather((k - (self.depth - 1)) % self.nbuf)
        return self.outs[b]

    def drain(self):
        """everything issued so far is complete (and gathered) once the current stream has reached this point"""
        self.ready(0)
        if self.exchanging and self.inline_gather is None:
            for j in range(max(0, self.steps - self.nbuf), self.steps):
                self._gather(j % self.nbuf)
        for b in range(self.nbuf):
            for w in self.pending[b]:
                if w is not None:
                    w.wait()
            self.pending[b] = []


def feed_sink(sink_decode_frame, chunks, masks, on_complete=None):
    """Rank-0 side: hand every delivered chunk to fountain_decoder_sink::decode_frame (fountain_decoder_sink.h:133-166) in
    frame order then chunk order -- exactly the order a single-threaded reference decoder would have produced.
    A positive return means "file <id> is complete": `on_complete(id)` runs at once (that is where the caller recovers /
    stores the file, which also marks it done so that later chunks of the same stream are ignored, as in
    fountain_decoder_sink.h:77-96,146-148). Returns the list of non-zero results."""
    out = []
    c = chunks.cpu().numpy().reshape(-1, modeb.CHUNKS_PER_FRAME, modeb.CHUNK)
    m = masks.cpu().numpy()
    for f in range(c.shape[0]):
        for j in range(modeb.CHUNKS_PER_FRAME):
            if int(m[f]) & (1 << j):
                r = sink_decode_frame(c[f, j])
                if r != 0:
                    out.append(r)
                if r > 0 and on_complete is not None:
                    on_complete(r)
    return out
