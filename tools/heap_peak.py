"""CPU: how many live priority-queue entries the reference's flood holds on camera-like frames (sizing of k_flood3's LDS heap).
Synthetic captures as bench.py's configs[4] row makes them (tools/extractbench.make_captures, torch on the CPU here), extracted and thresholded
by the oracle, then co_symbol_pass with its heap statistics. Test infrastructure: uses oracle/ only."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libcimbar_amd import framegen
from oracle import pyref
from tools import extractbench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = pyref.oracle_lib()
synth = framegen.FrameSynth("cpu")
frames = synth.frames_from_payload(framegen.synth_payload(n, seed=777))
caps = extractbench.make_captures(frames).numpy()
rows = []
for k in range(n):
    out = np.zeros((1024, 1024, 3), np.uint8)
    corners = np.zeros(8, np.float32)
    r = L.co_extract(pyref.P(np.ascontiguousarray(caps[k])), 1920, 1080, pyref.P(out), pyref.P(corners))
    plane = np.zeros(1024 * 128, np.uint8)
    L.co_threshold_bitplane(pyref.P(out), 1024, 1024, 1 if r == 2 else 0, pyref.P(plane))
    visit = np.zeros(4 * 12400, np.int32)
    cnt = L.co_symbol_pass(pyref.P(plane), pyref.P(visit), None)
    rows.append({"capture": k, "extract": r, "cells": cnt, "heap_peak": L.co_last_heap_peak(), "pops": L.co_last_heap_pops()})
    print(rows[-1], flush=True)
print(json.dumps({"max_peak": max(r["heap_peak"] for r in rows)}))
