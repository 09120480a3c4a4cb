"""tools/heap_model.cpp -- the CPU model of the exact flood replay's lane-parallel heap operations (csrc/k2b_flood.hip.inc: WaveHeap::push,
WaveHeap::pop with five heap levels per LDS round trip, WaveHeap7::pop7 with six, WaveHeap7::push4, and k_flood3's rule for which entry is on top after a
step's pushes) against libstdc++'s own std::push_heap / std::pop_heap, which is what the reference's std::priority_queue runs
(FloodDecodePositions.h:18-28,48). The kernels restate the model; tests/test_gpu_flood.py checks them against the oracle on the device."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("heapmodel") / "libheap_model.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "heap_model.cpp")], check=True)
    L = ctypes.CDLL(so)
    L.heap_model_fuzz.restype = ctypes.c_long
    L.heap_model_fuzz.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_long)]
    return L


# (seed, steps, priorities in use, percentage of big bursts, hover size): ties everywhere (1..3 priorities), camera-like alphabets (12..20),
# heaps that stay tiny and run empty, heaps around the block boundaries of both pops (31 / 63 / 2047 / 4095 entries), and ones that grow
# past 2^16 entries (three six-level rounds, four five-level ones)
CASES = [(1, 120000, 3, 60, 0), (2, 120000, 12, 30, 0), (3, 80000, 1, 80, 0), (4, 150000, 20, 40, 0), (5, 60000, 2, 95, 0), (6, 100000, 6, 50, 0),
         (7, 60000, 4, 0, 3), (8, 60000, 9, 0, 30), (9, 60000, 2, 0, 62), (10, 80000, 13, 0, 2046), (11, 80000, 3, 0, 4094), (12, 80000, 5, 0, 9000),
         (13, 60000, 1, 0, 64), (14, 60000, 64, 0, 500), (15, 200000, 7, 0, 32760), (16, 100000, 3, 0, 16380)]


# 5: WaveHeap::pop, 6: the model's first six-level pop (the kernels' pop7 replaced it; kept as a second witness), 7: pop7 (six levels, one write per path node), 8: pop7 + push4 (a burst's pushes four at a time, one gather and
# one scatter; seed 15 hovers where push4 hands over to the one-by-one push because an entry could have more than 14 ancestors)
@pytest.mark.parametrize("levels", [5, 6, 7, 8])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"seed{c[0]}-prio{c[2]}-hover{c[4]}")
def test_lane_parallel_heap_equals_libstdcxx(model, levels, case):
    seed, steps, prios, bias, hover = case
    mx = ctypes.c_long(0)
    bad = model.heap_model_fuzz(levels, seed, steps, prios, bias, hover, ctypes.byref(mx))
    assert bad == 0, f"step {bad}: the model's heap (or its predicted top) differs from std::push_heap / std::pop_heap"
    if hover:
        assert mx.value >= hover
    elif bias >= 50:
        assert mx.value > 65536          # deep enough for the last round of either pop
