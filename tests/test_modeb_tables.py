import numpy as np

from libcimbar_amd import framegen, modeb


def test_constants():
    # GridConf.h:121-142 / SURVEY.md section 2 table
    assert modeb.NCELLS == 112 * 112 - 4 * 36 == 12400
    assert modeb.NCELLS * 4 // 8 == 6200 and modeb.NCELLS * 2 // 8 == 3100
    assert 6200 // 155 == 40 and 3100 // 155 == 20
    assert (9300 * 125 // 155) // 12 == 625
    assert len(modeb.make_header(1, 0x1FFFFFF, 513)) == 6
    assert modeb.make_header(5, 0x1234567, 0x0102) == bytes([0x85, 0x23, 0x45, 0x67, 1, 2])


def test_positions_and_interleave_are_bijective():
    xy = modeb.cell_positions()
    assert len({tuple(p) for p in xy}) == modeb.NCELLS
    assert xy.min() == 8 and xy.max() == 1007
    idx = modeb.interleave_indices()
    assert sorted(idx.tolist()) == list(range(modeb.NCELLS))
    rev = modeb.interleave_reverse()
    assert (rev[idx] == np.arange(modeb.NCELLS)).all()
    # closed forms used by the kernels
    s = np.arange(modeb.NCELLS)
    p, sp = s // 6200, s % 6200
    assert (idx == 6200 * p + sp // 40 + 155 * (sp % 40)).all()


def test_synth_payload_headers():
    p = framegen.synth_payload(2, seed=1, encode_id=9, file_size=70000, first_block=65530).numpy().reshape(24, 625)
    assert (p[:, 0] == 9).all() and ((p[:, 1].astype(int) << 16) + (p[:, 2].astype(int) << 8) + p[:, 3] == 70000).all()
    ids = (p[:, 4].astype(int) << 8) + p[:, 5]
    assert list(ids[:8]) == [65530, 65531, 65532, 65533, 65534, 65535, 0, 1]


def test_classifier_division_can_be_done_in_float32():
    # CimbDecoder.cpp:180 divides in double and narrows; the HIP classifier divides in binary32. Double rounding
    # through binary64 is innocuous for a quotient of two binary32 numbers -- spot-check a few million denominators.
    g = np.random.default_rng(0)
    d = np.concatenate([np.arange(1, 256, dtype=np.float32), g.uniform(1e-3, 300, 2_000_000).astype(np.float32)])
    a = (np.float64(255.0) / d.astype(np.float64)).astype(np.float32)
    b = np.float32(255.0) / d
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
