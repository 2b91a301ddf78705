"""The approximate-arithmetic build's OWN bars (libspcbpt_hip_fast.so: csrc/Makefile).  NOT collected by the
default run (the name does not match test_*.py): tests/test_gpu_fast_build.py starts ONE child interpreter with
SPCBPT_LIB=<the fast library> that runs this file together with the image-level parity tests of the IEEE build, unchanged.

The fast library replaces correctly rounded FP32 division and square root by the hardware's reciprocal / square root (<= 2.5 / 2 ulp),
as the reference's own `--use_fast_math` build does (src/CMakeLists.txt:214).  Function-by-function comparisons against the oracle
(BSDF values within 1e-6, emitter radiance within 1e-4, film hashes) assume the oracle's operations and stay with the IEEE build.
What this build is held to:
  * it IS the fast library (spcbpt_build_arithmetic() == "approx": conftest.py checks SPCBPT_EXPECT_ARITHMETIC);
  * image-level parity against the oracle at the IEEE build's thresholds (test_gpu_parity.py image tests, run by the parent's list);
  * recursive-MIS weights = the balance heuristic, sum of the strategies' weights = 1 within 5e-4 for 99.5 % of the paths and within
    1e-4 in the median (test_gpu_first_principles.py + test_partition_median below);
  * SPCBPT and PT converge to the same image mean (below: 1.5 % at 96 spp on the Cornell box, the IEEE build's own bound);
  * the films of fixed seeds are dumped for the parent, which compares them pixel by pixel with the IEEE library's."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W = H = 96


def _renderer(pkg, scene, lt=(3000, 64, 1)):
    r = pkg.Renderer(scene, 0)
    cam = scene.camera
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    r.resize(W, H)
    r.set_light_trace(*lt)
    return r


def test_this_is_the_fast_library(gpu, pkg, hip_lib):
    assert hip_lib.spcbpt_build_arithmetic() == b"approx"
    assert os.path.samefile(os.environ["SPCBPT_LIB"], pkg.api.FAST_LIB_PATH)


def test_spcbpt_and_pt_agree_in_the_mean(gpu, pkg):
    scene = pkg.scenes.cornell_box()
    r = _renderer(pkg, scene)
    r.set_pretrace(20000, 10)
    r.preprocess(target_paths=100000, target_q_paths=100000, train=True)
    for f in range(96):
        r.render_frame("SPCBPT_eye", f)
    sp = r.read_accum()[..., :3]
    r.clear_accum()
    for f in range(96):
        r.launch("pt", f)
    pt = r.read_accum()[..., :3]
    assert np.isfinite(sp).all() and np.isfinite(pt).all()
    assert abs(sp.mean() - pt.mean()) / pt.mean() < 0.015, (sp.mean(), pt.mean())


def test_partition_median(gpu, pkg, ob):
    """Sum over the strategies of the device's recursive-MIS weights on explicit paths: median |sum - 1| < 1e-4."""
    from tests.test_gpu_first_principles import _weights
    scene = pkg.scenes.cornell_box()
    cam = scene.camera
    r, o = pkg.Renderer(scene, 0), ob.Oracle(scene)
    for x in (r, o):
        x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
        x.resize(64, 64)
        x.set_light_trace(3000, 64, 1)
    r.set_pretrace(20000, 10)
    r.preprocess(target_paths=100000, target_q_paths=100000, train=True)
    o.set_subspace(*r.get_subspace())
    r.launch("light trace", 1); r.build_sampler()
    for depth in (2, 3):
        w, truth, ev, lv = o.quad_partition(depth, 600, vertices=True)
        ok = np.abs(w[:, 0] - 1) < 1e-3
        truth, ev, lv = truth[ok], ev[ok], lv[ok]
        total = truth[:, 0].astype(np.float64)
        for k in range(depth):
            total += _weights(r, ev[:, k], lv[:, k])
        assert np.median(np.abs(total - 1)) < 1e-4, (depth, np.median(np.abs(total - 1)))


def test_dump_films_for_the_parent(gpu, pkg):
    out = os.environ.get("SPCBPT_FAST_DUMP")
    if not out:
        pytest.skip("no SPCBPT_FAST_DUMP")
    scene = pkg.scenes.cornell_box()
    r = _renderer(pkg, scene)
    r.set_subspace()
    for f in range(4):
        r.render_frame("SPCBPT_eye", f)
    sp = r.read_accum().copy()
    r.clear_accum()
    for f in range(4):
        r.launch("pt", f)
    np.savez(out, spcbpt=sp, pt=r.read_accum())
