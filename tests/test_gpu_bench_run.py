"""bench.py end to end at a reduced size (the driver runs it unattended: a flag combination that trips the host loop's own
consistency checks -- the build-ahead bookkeeping, uneven launches -- must show up here, not there)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "320", "--height", "184", "--tris", "20000", "--light-paths", "4000", "--tuple", "minimal", "--no-cpu-baseline"]


def _run(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, (args, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (args, p.stdout[-2000:])     # the contract: ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("args", [
    ["--steps", "20", "--warmup", "5"],                                                            # the driver's flags, every default pass (256-step run, sync-each, viewer)
    ["--steps", "5", "--warmup", "2", "--long-steps", "6", "--sync-each-frames", "2"],            # one launch per phase, every extra pass on
    ["--steps", "33", "--warmup", "3", "--long-steps", "0", "--sync-each-frames", "0"],           # two uneven launches (17 + 16), builds a batch ahead
    ["--steps", "33", "--warmup", "3", "--long-steps", "34", "--sync-each-frames", "0", "--build-ahead", "0"],
    ["--steps", "4", "--warmup", "1", "--long-steps", "0", "--sync-each-frames", "0", "--eye-batch", "1"],   # one eye launch per frame
    ["--steps", "6", "--warmup", "2", "--long-steps", "0", "--sync-each-frames", "0", "--force-exchange"],   # the RCCL path at world size 1
    ["--steps", "33", "--warmup", "3", "--long-steps", "40", "--sync-each-frames", "0", "--force-exchange"],  # ... with uneven launches, exchanges a batch ahead
], ids=["driver-form", "all-passes", "uneven-launches", "no-build-ahead", "unbatched", "forced-exchange", "forced-exchange-uneven"])
def test_bench_line(gpu, args):
    d = _run(args)
    steps = int(args[args.index("--steps") + 1])
    assert d["metric"].startswith("Mpaths/sec") and d["unit"] == "Mpaths/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["unit"] == "GB/s" and r["achieved"] > 0 and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["config"]["workload"] and d["dtype"] == "f32" and d["data"] == "synthetic"
    ev = d["events_per_eye_path"]
    assert ev["closest_rays"] >= 1.0 and ev["node_visits"] > 1.0


def test_forced_exchange_renders_the_same_film_with_and_without_builds_ahead(gpu, tmp_path):
    """The sharded job's loop (RCCL path at world size 1): exchanging and building a batch ahead changes when the calls are queued,
    not what any frame is rendered from."""
    import numpy as np
    films = []
    for ahead in ("1", "0"):
        out = str(tmp_path / f"film_{ahead}.npy")
        d = _run(["--steps", "20", "--warmup", "5", "--long-steps", "40", "--sync-each-frames", "0", "--force-exchange", "--build-ahead", ahead,
                  "--write-image", out])
        assert d["config"]["sampler_builds_ahead"] == (20 if ahead == "1" else 0)
        films.append(np.load(out))
    assert films[0].shape == films[1].shape and np.isfinite(films[0]).all() and films[0][..., :3].mean() > 0.01
    assert np.array_equal(films[0], films[1])
