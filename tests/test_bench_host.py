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


def test_cpu_baseline_runs_on_the_cpus_the_box_grants(tmp_path):
    """bench.py: host_cpus() = min(os.cpu_count(), affinity mask, cgroup quota).  A one-GPU box of the pool shows its host's 256 hardware
    threads and grants 16 CPUs (cpu.max = "1600000 100000", profiles/r06_hostcpu.txt): 256 oracle threads there ran like 11."""
    b = _bench()
    n = os.cpu_count() or 1
    assert 1 <= b.host_cpus() <= n
    v2 = tmp_path / "v2"; v2.mkdir()
    (v2 / "cpu.max").write_text("200000 100000\n")                 # cgroup v2: two CPUs
    assert b.host_cpus(str(v2)) == min(2, n, len(os.sched_getaffinity(0)))
    (v2 / "cpu.max").write_text("150000 100000\n")                 # a fractional quota rounds up
    assert b.host_cpus(str(v2)) == min(2, n, len(os.sched_getaffinity(0)))
    (v2 / "cpu.max").write_text("max 100000\n")                    # no quota
    assert b.host_cpus(str(v2)) == min(n, len(os.sched_getaffinity(0)))
    v1 = tmp_path / "v1"; (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("100000\n"); (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")   # cgroup v1: one CPU
    assert b.host_cpus(str(v1)) == 1
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")           # v1: no quota
    assert b.host_cpus(str(v1)) == min(n, len(os.sched_getaffinity(0)))
    assert b.host_cpus(str(tmp_path / "nothing")) == min(n, len(os.sched_getaffinity(0)))
