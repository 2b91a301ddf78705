"""The opt-in approximate-arithmetic library (libspcbpt_hip_fast.so) against bars of its own -- see tests/fast_build_bars.py for what
they are and why the function-level tests stay with the IEEE build.  Two libraries with the same exported names cannot live in one
process, so the fast one is exercised in ONE child interpreter (one more process on the GPU, within the box's limit)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.parity_util import image_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the image-level tests of the IEEE build, run UNCHANGED (same thresholds) on the fast library, + its own bars
CHILD_TESTS = [
    "tests/fast_build_bars.py",
    "tests/test_gpu_parity.py::test_pt_image_matches_oracle",
    "tests/test_gpu_parity.py::test_spcbpt_image_matches_oracle",
    "tests/test_gpu_parity.py::test_spcbpt_with_multi_leaf_trees_and_textures",
    "tests/test_gpu_first_principles.py",
    "tests/test_gpu_scene_file.py",
]


def test_fast_library_meets_its_image_level_bars(gpu, pkg, tmp_path):
    assert os.path.exists(pkg.api.FAST_LIB_PATH), "libspcbpt_hip_fast.so not built (make -C spcbpt-optix7_amd/csrc)"
    assert pkg.load_library().spcbpt_build_arithmetic() == b"ieee"   # this process tests the default library
    dump = str(tmp_path / "fast_films.npz")
    env = dict(os.environ, SPCBPT_LIB=pkg.api.FAST_LIB_PATH, SPCBPT_EXPECT_ARITHMETIC="approx", SPCBPT_FAST_DUMP=dump)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + CHILD_TESTS, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    print(p.stdout[-3000:])
    assert p.returncode == 0, p.stdout[-3000:]
    # the same frames from the IEEE library in this process: the two builds differ in last bits that grow along a path; pixel by pixel
    # they agree like the device agrees with the oracle (>= 99 % of the pixels within 2e-3 relative), and their means within 2e-3
    fast = np.load(dump)
    scene = pkg.scenes.cornell_box()
    r = pkg.Renderer(scene, 0)
    cam = scene.camera
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
    r.resize(96, 96)
    r.set_light_trace(3000, 64, 1)
    r.set_subspace()
    for f in range(4):
        r.render_frame("SPCBPT_eye", f)
    s = image_parity(fast["spcbpt"][..., :3], r.read_accum()[..., :3])
    assert s["frac_close"] >= 0.99 and s["mean_rel"] < 2e-3, s
    r.clear_accum()
    for f in range(4):
        r.launch("pt", f)
    s = image_parity(fast["pt"][..., :3], r.read_accum()[..., :3])
    assert s["frac_close"] >= 0.99 and s["mean_rel"] < 2e-3, s
