"""The N>1 path on CPU: world_size 2 and 3 over gloo (torch.distributed.run on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_frame_decomposition_over_gloo(ob, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:]


def test_partition_helpers(pkg):
    d = pkg.dist
    for m, w in ((100000, 8), (1201, 3), (7, 8)):
        cover = []
        for r in range(w):
            b, c = d.core_range(m, r, w)
            cover += list(range(b, b + c))
        assert cover == list(range(m))
    for h, w in ((1080, 8), (56, 3), (8, 2)):
        rows = sorted(y for r in range(w) for y in d.rows_of_rank(h, r, w))
        assert rows == list(range(h))
        for r in range(w):
            b, e, s = d.band_rows(h, r, w)
            assert b == 8 * r and e == h and s == w
