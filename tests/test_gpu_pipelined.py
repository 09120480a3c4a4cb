"""GPU: the pipelined entry point (threshold pass of batch k+1 overlapping the rest of batch k) returns exactly what the ordinary
one does for the same sequence of batches, including the colour-correction carry from batch to batch and the flood pass."""
import numpy as np
import pytest
import torch

from libcimbar_amd import decoder as D
from libcimbar_amd import modeb
from tests import frames as F

pytestmark = pytest.mark.gpu


def test_pipelined_sequence_equals_ordinary_sequence(synth):
    dev = torch.device("cuda", 0)
    payload, clean = F.clean_frames(synth, 6, seed=77)
    blank = np.zeros_like(clean[0])
    # batch 1 ends with a frame that has no matrix of its own; batch 2 starts with one: it must inherit batch 1's last matrix
    batches = [
        [clean[0], F.add_noise(clean[1], 40, 1), F.shift(clean[2], 2, 1)],
        [blank, F.add_noise(clean[3], 60, 2), clean[4]],
        [blank, F.shift(clean[5], -1, 2), F.add_noise(clean[0], 25, 3)],
        [F.add_noise(clean[2], 90, 4), blank, blank],
    ]
    tens = [torch.from_numpy(np.ascontiguousarray(np.stack(b))).to(dev) for b in batches]
    st = torch.cuda.current_stream(dev).cuda_stream

    def run(pipelined):
        dec = D.HipDecoder(0)
        outs = [(torch.zeros((3, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros((3,), dtype=torch.int32, device=dev)) for _ in tens]
        cols = []
        for t, (c, m) in zip(tens, outs):
            if pipelined:
                dec.decode_batch_pipelined(t.data_ptr(), 3, c.data_ptr(), m.data_ptr(), False, 2, st)
            else:
                dec.decode_batch_device(t.data_ptr(), 3, c.data_ptr(), m.data_ptr(), False, 2, st)
                torch.cuda.synchronize()
                cols.append(dec.tap(D.TAP_COLORS, 3).copy())
        if pipelined:
            dec.pipeline_wait(st)
        torch.cuda.synchronize()
        res = [(c.cpu().numpy(), m.cpu().numpy()) for c, m in outs]
        ccm = dec.get_ccm()
        dec.close()
        return res, ccm, cols

    want, wccm, _ = run(False)
    got, gccm, _ = run(True)
    for k, ((wc, wm), (gc, gm)) in enumerate(zip(want, got)):
        assert (wm == gm).all(), f"batch {k}: masks {wm} vs {gm}"
        assert (wc == gc).all(), f"batch {k}: chunk bytes differ"
    assert wccm[0] == gccm[0] and wccm[1].tobytes() == gccm[1].tobytes()


def test_pipelined_then_ordinary_calls_mix(synth):
    dev = torch.device("cuda", 0)
    payload, clean = F.clean_frames(synth, 2, seed=5)
    t = torch.from_numpy(np.ascontiguousarray(clean)).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    dec = D.HipDecoder(0)
    c = [torch.zeros((2, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev) for _ in range(3)]
    m = [torch.zeros((2,), dtype=torch.int32, device=dev) for _ in range(3)]
    dec.decode_batch_pipelined(t.data_ptr(), 2, c[0].data_ptr(), m[0].data_ptr(), False, 2, st)
    dec.decode_batch_pipelined(t.data_ptr(), 2, c[1].data_ptr(), m[1].data_ptr(), False, 2, st)
    dec.decode_batch_device(t.data_ptr(), 2, c[2].data_ptr(), m[2].data_ptr(), False, 2, st)   # waits for the pipeline by itself
    torch.cuda.synchronize()
    for k in range(3):
        assert bool((m[k] == 0xFFF).all()) and (c[k].cpu().numpy() == payload).all()
    total, chunks, masks = dec.decode_batch(clean)   # host entry point after pipelined use
    assert total == 15000 and (chunks.reshape(2, -1) == payload).all()


def test_pipeline_keeps_order_over_many_batches(synth):
    """more batches than the pipeline is deep, consumed D-1 steps behind: every scratch set and stream is reused several times"""
    dev = torch.device("cuda", 0)
    payload, clean = F.clean_frames(synth, 4, seed=21)
    variants = [np.ascontiguousarray(np.stack([clean[k], F.add_noise(clean[(k + 1) % 4], 35 + 10 * k, k), F.shift(clean[(k + 2) % 4], k - 1, 1)]))
                for k in range(4)]
    tens = [torch.from_numpy(v).to(dev) for v in variants]
    st = torch.cuda.current_stream(dev).cuda_stream
    nb = 11
    ref_dec = D.HipDecoder(0)
    want = []
    for k in range(nb):
        want.append(ref_dec.decode_batch(variants[k % 4]))
    ref_dec.close()

    dec = D.HipDecoder(0)
    depth = dec.pipeline_depth
    assert 2 <= depth <= 4
    outs = [(torch.zeros((3, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev), torch.zeros((3,), dtype=torch.int32, device=dev)) for _ in range(nb)]
    got = [None] * nb
    for k in range(nb):
        c, m = outs[k]
        dec.decode_batch_pipelined(tens[k % 4].data_ptr(), 3, c.data_ptr(), m.data_ptr(), False, 2, st)
        j = k - (depth - 1)
        if j >= 0:
            dec.pipeline_wait(st, keep_newest=depth - 1)
            got[j] = (outs[j][0].clone(), outs[j][1].clone())      # stream-ordered copies: must already see batch j's results
    dec.pipeline_wait(st)
    torch.cuda.synchronize()
    for j in range(nb):
        c, m = got[j] if got[j] is not None else outs[j]
        assert (m.cpu().numpy().astype(np.uint32) == want[j][2]).all(), j
        assert (c.cpu().numpy().reshape(3, 12, 625) == want[j][1]).all(), j
    dec.close()


def test_small_pipelined_batches_then_a_larger_host_output_batch(synth):
    """the host-output staging buffers are shared by every scratch set: pipelined calls with small n on the other sets must not
    shrink what a later, larger host-output batch on the first set writes into (round-1 advisor finding)"""
    dev = torch.device("cuda", 0)
    payload, clean = F.clean_frames(synth, 8, seed=99)
    st = torch.cuda.current_stream(dev).cuda_stream
    dec = D.HipDecoder(0)
    total, chunks, masks = dec.decode_batch(clean)                       # set 0: capacity 8, staging 8 frames
    assert total == 8 * 7500
    t = torch.from_numpy(np.ascontiguousarray(clean[:2])).to(dev)
    c = [torch.zeros((2, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev) for _ in range(dec.pipeline_depth)]
    m = [torch.zeros((2,), dtype=torch.int32, device=dev) for _ in range(dec.pipeline_depth)]
    for k in range(dec.pipeline_depth):                                  # rotates through every set and back to set 0
        dec.decode_batch_pipelined(t.data_ptr(), 2, c[k].data_ptr(), m[k].data_ptr(), False, 2, st)
    dec.pipeline_wait(st)
    torch.cuda.synchronize()
    total, chunks, masks = dec.decode_batch(clean)
    assert total == 8 * 7500 and (masks == 0xFFF).all() and (chunks.reshape(8, -1) == payload).all()
    rc, data, ok = dec.decode_plain_batch(clean)
    assert rc == 8 * 7500 and ok.all() and (data == payload).all()
    dec.close()


def test_batch_without_any_matrix_takes_the_carried_one(synth):
    """a long batch whose frames yield no matrix of their own (blank frames) resolves the matrix in force by the workgroup-wide
    backward probe of k_colors: every frame must use the carried matrix, and the carry must survive the batch"""
    dev = torch.device("cuda", 0)
    payload, clean = F.clean_frames(synth, 1, seed=3)
    dec = D.HipDecoder(0)
    dec.decode_batch(clean)
    active, carried = dec.get_ccm()
    assert active
    n = 700
    blank = torch.zeros((n, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
    blank[n - 1] = torch.from_numpy(clean[0]).to(dev)
    c = torch.zeros((n, modeb.FRAME_BYTES), dtype=torch.uint8, device=dev)
    m = torch.zeros((n,), dtype=torch.int32, device=dev)
    dec.decode_batch_device(blank.data_ptr(), n, c.data_ptr(), m.data_ptr(), False, 2, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    used = dec.tap(D.TAP_CCM, n)
    assert (used[:n - 1, :9].tobytes() == np.tile(carried.reshape(1, 9), (n - 1, 1)).astype(np.float32).tobytes())
    assert (used[:n - 1, 9] != 0).all()
    assert int(m[n - 1].item()) == 0xFFF and (c[n - 1].cpu().numpy() == payload[0]).all()
    dec.close()
