"""BASELINE config 4 at its stated length through an engine sized for what is in flight (ocr_amd/stencil_window.py):
the 5-point stencil issued as a sliding window over time, ocrgpu_db_destroy + ocrgpu_reclaim behind the firing front
(the reference frees as it goes: destructTaskHc hc-task.c:721-790, destructEventHc hc-event.c:185-242).

Parity: (1) on a 64 x 64 x 1000 twin the record the windows leave -- level and HASH value of all 4 096 001 EDTs, and the
running digest -- equals the CPU restatement of the SAME program written down whole (opstream.stencil_dag); (2) at sizes
the restatement is not run on, every tile-step's level is the closed form t - (first step of its window) and every value
equals a vectorised restatement of the HASH recurrence, itself held to the CPU restatement by (1)."""
import numpy as np
import pytest

from ocr_amd import opstream as ops
from ocr_amd.stencil_window import StencilWindow, record_digest, t_mix64
from oracle import ref as O

pytestmark = pytest.mark.gpu


def _run(gx, gy, steps, window, **kw):
    w = StencilWindow(gx, gy, steps, window, **kw).run()
    s = w.summary()
    assert s["level_errors"] == 0 and s["value_errors"] == 0, s
    assert sum(s["err"]) == 0, s
    assert s["fired_total"] == w.N + 1, s
    return w, s


def test_mix64_on_int64_tensors_is_the_unsigned_mix():
    import torch
    x = ops.splitmix64(99, 4096)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        want = z ^ (z >> np.uint64(31))
    got = t_mix64(torch.from_numpy(x.view(np.int64)).cuda()).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)


def test_twin_1000_steps_equals_the_restatement_of_the_whole_program():
    G, T, W = 64, 1000, 24
    w, s = _run(G, G, T, W, keep_record=True)
    assert s["windows"] == -(-(T - 1) // W) or s["windows"] == -(-T // W), s
    assert s["engine_max_edt"] < (W + 3) * G * G                      # sized for the window, not for 1000 steps
    want = O.run_restatement(ops.stencil_dag(G, T))
    N = G * G * T
    assert np.array_equal(w.record_level[:N], want.level[:N])
    assert np.array_equal(w.record_val[:N], want.val[:N])
    assert w.record_level[N] == want.level[N] == T                    # the sink: one level behind the last step's latch check-outs
    assert w.digest == record_digest(want.level, want.val, N)


@pytest.mark.parametrize("gx,gy,steps,window", [(16, 16, 10, 3), (16, 16, 10, 1), (16, 16, 10, 64), (8, 32, 33, 8), (1, 5, 7, 2)])
def test_small_shapes_against_the_restatement(gx, gy, steps, window):
    w, s = _run(gx, gy, steps, window, keep_record=True)
    if gx == gy:
        want = O.run_restatement(ops.stencil_dag(gx, steps))
        N = gx * gy * steps
        assert np.array_equal(w.record_level[:N + 1], want.level[:N + 1])
        assert np.array_equal(w.record_val[:N], want.val[:N])


def test_medium_grid_every_level_and_value():
    # 512 x 256 tiles x 64 steps = 8.4M tile-steps through an engine of 10 steps
    w, s = _run(512, 256, 64, 8)
    assert s["engine_max_edt"] == 10 * 512 * 256 + 8
    print(s)
