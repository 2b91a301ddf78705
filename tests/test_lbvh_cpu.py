"""Structural check of the host BVH builders (csrc/lbvh.cpp) without a GPU: tests/native/lbvh_check.cpp is compiled with
g++ and verifies that every triangle sits in exactly one leaf, that every decoded (8-bit quantised) child box contains
its whole subtree, and that empty slots trail the used ones — for the SAH builder and the Morton-order LBVH."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lbvh") / "lbvh_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "lbvh_check.cpp")], check=True)
    return exe


@pytest.mark.parametrize("builder", ["sah", "lbvh", "lbvh-sah"])
@pytest.mark.parametrize("n,seed", [(1, 1), (4, 1), (5, 2), (777, 3), (60000, 4)])
def test_bvh_structure(checker, builder, n, seed):
    env = dict(os.environ, SPCBPT_BVH=builder)
    r = subprocess.run([checker, str(n), str(seed)], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_offline_bvh_evaluator_builds_and_traverses(tmp_path, pkg):
    """tools/bvh_eval.cpp (developer tool: the product's builder + a CPU traversal of the quantised 4-wide nodes with the device's rules)
    on a small scene: every closest-hit ray from inside the closed Cornell box hits something, node visits are sane."""
    import numpy as np
    exe = str(tmp_path / "bvh_eval")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "bvh_eval.cpp")], check=True)
    scene = pkg.scenes.bedroom(target_tris=20000, tex_size=8)
    V = np.ascontiguousarray(scene.vertices, np.float32)
    I = np.ascontiguousarray(scene.indices, np.uint32)
    mesh = tmp_path / "mesh.bin"
    with open(mesh, "wb") as f:
        f.write(np.array([V.shape[0], I.shape[0]], np.int32).tobytes()); f.write(V.tobytes()); f.write(I.tobytes())
    r = subprocess.run([exe, str(mesh), "20000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    line = next(l for l in lines if l.startswith("closest:"))
    closest = dict(zip(("visits", "leaves", "tris", "hit"), [float(x) for x in __import__("re").findall(r"[-+]?\d*\.\d+", line)]))
    steps = [float(x) for x in __import__("re").findall(r"[-+]?\d*\.\d+", next(l for l in lines if l.startswith("steps per ray")))]
    assert steps[0] <= closest["visits"] + closest["tris"] + 1e-6 and steps[0] >= closest["visits"]   # a fan pair is ONE triangle step (round 6)
    assert 3 < closest["visits"] < 40 and closest["tris"] < 10 and closest["hit"] > 0.4, r.stdout
