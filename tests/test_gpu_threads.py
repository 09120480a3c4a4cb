"""GPU: one context per host thread, several threads at once -- the reference's model (Config and the CCM are thread_local,
SURVEY 8(b) "Threading"): contexts must not share any mutable state."""
import threading

import numpy as np
import pytest

from libcimbar_amd import decoder as D
from tests import frames as F

pytestmark = pytest.mark.gpu


def test_contexts_are_independent_across_threads(synth):
    payload, frames = F.clean_frames(synth, 6, seed=404)
    sets = []
    for t in range(3):   # each thread gets its own mix: clean, noisy (own CCM), shifted (flood pass)
        fr = [frames[2 * t], F.add_noise(frames[2 * t + 1], 30 + 20 * t, t), F.shift(frames[2 * t], 1 + t, -t)]
        sets.append(np.ascontiguousarray(np.stack(fr)))
    serial = []
    for s in sets:
        d = D.HipDecoder(0)
        serial.append(d.decode_batch(s))
        d.close()
    results = [None] * 3
    errors = []

    def work(t):
        try:
            d = D.HipDecoder(0)
            for _ in range(4):
                results[t] = d.decode_batch(sets[t])
                d.reset_ccm()
            d.close()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for t in range(3):
        assert results[t][0] == serial[t][0]
        assert (results[t][1] == serial[t][1]).all() and (results[t][2] == serial[t][2]).all()
