"""Test helper: a batch of mode-B frames through the reference build's Decoder::decode_fountain (oracle/_ref) on a thread pool, with the semantics of ONE
reference thread decoding the frames in order (what a GPU batch call reproduces)."""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from libcimbar_amd import modeb
from oracle import pyref


def reference_batch(frames_host, preprocess=0, cc=2, threads=None):
    """What ONE reference thread makes of the batch, frame 0 .. n-1 in order (the semantics of a GPU batch: the colour-correction matrix is carried from
    frame to frame, CimbDecoder.cpp:69-73 `static thread_local`), computed on a pool: thread t decodes a contiguous slice in order after priming its
    thread_local matrix with the slice's predecessor frame. That equals the sequential run whenever the predecessor derives a matrix of its own -- every
    frame whose symbol stream decodes does (CimbReader.cpp:169-267), which holds for the clean and the RS-correctable batches compared here and is
    asserted (full byte count) frame by frame."""
    ref = pyref.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    n = frames_host.shape[0]
    threads = max(1, min(threads or len(os.sched_getaffinity(0)), 16, n))
    per = (n + threads - 1) // threads
    chunks = np.zeros((n, modeb.FRAME_BYTES), np.uint8)
    masks = np.zeros(n, np.uint32)

    def work(t):
        lo, hi = t * per, min(n, (t + 1) * per)
        if lo >= hi:
            return
        ref.ref_configure(68)                       # (Config is thread_local as well)
        scratch = np.zeros(modeb.FRAME_BYTES, np.uint8)
        m = ctypes.c_uint32(0)
        if lo > 0:
            r = ref.ref_decode_fountain(pyref.P(frames_host[lo - 1]), modeb.IMG, modeb.IMG, preprocess, cc, 1, pyref.P(scratch), ctypes.byref(m))
            assert r == modeb.FRAME_BYTES            # the predecessor made its own matrix: the primed state is the sequential one
        for k in range(lo, hi):
            r = ref.ref_decode_fountain(pyref.P(frames_host[k]), modeb.IMG, modeb.IMG, preprocess, cc, 1 if k == 0 else 0, pyref.P(chunks[k]), ctypes.byref(m))
            assert r == modeb.FRAME_BYTES
            masks[k] = m.value

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(work, range(threads)))
    return chunks, masks


