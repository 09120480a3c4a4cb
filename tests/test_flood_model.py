"""CPU: the claim behind the batch-parallel flood (k_flood_wave, libcimbar_amd/csrc/k2c_floodwave.hip.inc), checked on its CPU model
oracle/flood_model.c: whenever the model CERTIFIES a frame (no rule B1..B6 fired), its symbols and drifted positions equal what the
reference's sequential priority flood produces (co_symbol_pass, itself pinned to the reference build in tests/test_oracle_vs_ref.py).
Also: it certifies the frames it is meant for (clean shifts, wiped regions) and declines noisy / resampled ones."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref
from oracle.pyref import P
from tests import frames as F

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), "oracle")
SO = os.path.join(ORACLE, "libflood_model.so")


@pytest.fixture(scope="module")
def model():
    srcs = [os.path.join(ORACLE, "flood_model.c"), os.path.join(ORACLE, "cimbar_oracle_extract.c")]
    deps = srcs + [os.path.join(ORACLE, "cimbar_oracle.c"), os.path.join(ORACLE, "cimbar_oracle.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(p) for p in deps):
        subprocess.run(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-std=gnu11", "-shared", "-o", SO, *srcs, "-lm"], check=True)
    return ctypes.CDLL(SO)


def flood_both(model, frame, prefix=64):
    plane = np.zeros(131072, np.uint8)
    model.co_threshold_bitplane(P(np.ascontiguousarray(frame)), 1024, 1024, 0, P(plane))
    visit = np.zeros(4 * 12400, np.int32)
    n = model.co_symbol_pass(P(plane), P(visit), None)
    v = visit.reshape(-1, 4)[:n]
    wsym = np.zeros(12400, np.uint8)
    wpos = np.zeros((12400, 2), np.int32)
    wsym[v[:, 0]] = v[:, 3]
    wpos[v[:, 0], 0] = v[:, 1]
    wpos[v[:, 0], 1] = v[:, 2]
    sym = np.zeros(12400, np.uint8)
    pos = np.zeros((12400, 2), np.int32)
    st = np.zeros(8, np.int32)
    rc = model.fm_flood(P(plane), prefix, P(sym), P(pos), P(st))
    return rc, bool((sym == wsym).all() and (pos == wpos).all()), st


def test_certified_frames_equal_the_sequential_flood(model, synth):
    payload, clean = F.clean_frames(synth, 6, seed=123)
    g = np.random.default_rng(11)
    certified = 0
    for it in range(36):
        fr = clean[it % 6]
        kind = it % 6
        if kind == 0:
            fr = F.shift(fr, int(g.integers(-3, 4)), int(g.integers(-3, 4)))
        elif kind == 1:
            fr = F.add_noise(F.shift(fr, int(g.integers(-2, 3)), int(g.integers(-2, 3))), int(g.integers(5, 40)), it)
        elif kind == 2:
            y0, x0 = int(g.integers(0, 900)), int(g.integers(0, 900))
            fr = F.blank_region(F.shift(fr, int(g.integers(-2, 3)), int(g.integers(-2, 3))), y0, y0 + int(g.integers(20, 400)), x0,
                                x0 + int(g.integers(20, 400)), value=int(g.integers(0, 256)))
        elif kind == 3:
            fr = F.add_noise(fr, int(g.integers(10, 80)), it)
        elif kind == 4:
            cut = int(g.integers(200, 800))
            fr = np.concatenate([F.shift(fr, 1, 0)[:cut], F.shift(fr, 0, 1)[cut:]], 0)     # a tear: two different shifts
        else:
            fr = F.shift(fr, int(g.integers(-1, 2)), int(g.integers(-1, 2))).copy()
            for _ in range(int(g.integers(1, 30))):
                y, x = int(g.integers(8, 1000)), int(g.integers(8, 1000))
                fr[y:y + 9, x:x + 9] = g.integers(0, 256, (9, 9, 3))
        for prefix in (32, 64):
            rc, same, st = flood_both(model, fr, prefix)
            if rc == 0:
                certified += 1
                assert same, f"case {it} kind {kind} prefix {prefix}: certified but different from the sequential flood"
    assert certified >= 20


def test_model_certifies_what_it_is_meant_for(model, synth):
    payload, clean = F.clean_frames(synth, 4, seed=77)
    for dy, dx in ((2, 1), (1, 0), (-1, -1), (0, 3)):
        rc, same, st = flood_both(model, F.shift(clean[0], dy, dx))
        assert rc == 0 and same, (dy, dx, st)
        assert st[1] <= 4 and st[2] <= 120          # a handful of super-rounds, tens of levels
    rc, same, _ = flood_both(model, F.blank_region(clean[1], 300, 420, 0, 1024))
    assert rc == 0 and same
    rc, _, _ = flood_both(model, F.add_noise(clean[2], 120, 2))
    assert rc != 0                                   # mixed priorities everywhere: left to the exact replay
    rc, _, _ = flood_both(model, F.rescale(clean[3], 6))
    assert rc != 0
