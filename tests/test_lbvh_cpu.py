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
