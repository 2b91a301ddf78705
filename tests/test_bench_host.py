"""Host-side arithmetic of bench.py (no GPU): how a run's steps are grouped into launches."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["bench_module"] = m
    spec.loader.exec_module(m)
    return m


def test_a_run_is_cut_into_few_equal_launches():
    b = _bench()
    assert b.frames_per_launch(20) == 20            # the driver's run: one launch
    assert b.frames_per_launch(32) == 32 and b.frames_per_launch(33) == 17 and b.frames_per_launch(64) == 32
    assert b.frames_per_launch(1) == 1 and b.frames_per_launch(5) == 5
    for k in range(1, 200):
        f = b.frames_per_launch(k)
        launches = -(-k // f)
        assert 1 <= f <= 32 and launches == max(1, -(-k // 32))          # never more launches than 32-frame batches need
        assert k - (launches - 1) * f > f - launches                      # the last launch is at most launches - 1 frames short
