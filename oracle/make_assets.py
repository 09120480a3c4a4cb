"""Regenerates fixtures that need the REFERENCE build (oracle/_ref/libcimbar_ref.so, i.e. /root/reference at build time):

  libcimbar_amd/data/modeb_template.npz   background + anchors + guides of an empty mode-B frame (CimbWriter.cpp:39-77), and
  libcimbar_amd/data/modebm_template.npz  the same for mode Bm (67, 1024x720),
                                          stored sparse (flat byte index, value) -- input of libcimbar_amd/framegen.py
  tests/golden/*.npz                      see tests/golden/README.md

Run here (in the build container); the outputs are committed so that nothing at test/bench time needs /root/reference.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyref import P, ref_lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    L = ref_lib()
    if L is None:
        raise SystemExit("oracle/_ref/libcimbar_ref.so missing: run `make -C oracle ref` first")
    os.makedirs(os.path.join(ROOT, "libcimbar_amd", "data"), exist_ok=True)
    for mode, (w, h), name in ((68, (1024, 1024), "modeb_template.npz"), (67, (1024, 720), "modebm_template.npz"), (66, (736, 637), "modebu_template.npz")):
        L.ref_configure(mode)
        t = np.zeros(w * h * 3, np.uint8)
        L.ref_template_frame(P(t))
        idx = np.nonzero(t)[0].astype(np.uint32)
        np.savez_compressed(os.path.join(ROOT, "libcimbar_amd", "data", name), idx=idx, val=t[idx])
        print("template mode %d: %d non-zero bytes" % (mode, idx.size))
    L.ref_configure(68)

if __name__ == "__main__":
    main()
