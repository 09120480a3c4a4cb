import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyref
    return pyref.oracle_lib()


@pytest.fixture(scope="session")
def ref():
    """The reference build (oracle/_ref). Tests that need it are skipped where it was never built."""
    from oracle import pyref
    L = pyref.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref/libcimbar_ref.so not built (needs /root/reference at build time)")
    return L


@pytest.fixture(scope="session")
def synth():
    from libcimbar_amd import framegen
    return framegen.FrameSynth("cpu")


@pytest.fixture(scope="session")
def hip_decoder():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from libcimbar_amd import HipDecoder
    return HipDecoder(0)
