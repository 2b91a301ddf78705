"""The tables the eye megakernel samples through (round 5: csrc/layout.h KParams::guide, cmf_guide1, gamma_q), read back through
spcbpt_debug_read_sampling_tables and checked against their definitions and against the reference's bisection (cuProg.h:245-264)
restated here.  The guided search must return the bisection's bin for every random number; that holds if every guide entry is a
lower bound of the bin for all the numbers of its bucket -- which is what these tests establish, next to the film hashes of
test_gpu_film_golden.py, which pin the end result."""
import os

import numpy as np
import pytest

from tests.parity_util import grid_tree_tuple

pytestmark = pytest.mark.gpu


def reference_bisection(cmf, u):
    """binary_sample (cuProg.h:245-264) as written: the bin and the probes it took."""
    size = len(cmf)
    mid, lo, hi = size // 2 - 1, 0, size
    while hi - lo > 1:
        if u < cmf[mid]:
            hi = mid + 1
        else:
            lo = mid + 1
        mid = (lo + hi) // 2 - 1
    return lo


def _setup(pkg, ob, lt=(4000, 64, 1)):
    scene = pkg.scenes.bedroom(target_tris=20000, tex_size=32)
    r = pkg.Renderer(scene, 0)
    o = ob.Oracle(scene)
    cam = scene.camera
    for x in (r, o):
        x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 4 / 3)
        x.resize(32, 24)
        x.set_light_trace(*lt)
    tup = grid_tree_tuple(pkg, o, scene)
    r.set_subspace(*tup)
    return scene, r, o, tup


def test_second_stage_guide_is_its_definition_and_a_lower_bound(gpu, pkg, ob):
    scene, r, o, tup = _setup(pkg, ob)
    r.launch("light trace", 7)
    r.build_sampler()
    sub, cmfs, jump, vc, pc = r.sampler_read()
    guide2, _, _ = r.sampling_tables(vc)
    assert vc > 1000 and (sub["size"] > 50).sum() >= 3
    rng = np.random.default_rng(5)
    checked = 0
    for s in sub[sub["size"] > 0]:
        b, n = int(s["jump_bias"]), int(s["size"])
        cmf = cmfs[b:b + n]
        assert cmf[-1] == 1.0 and (np.diff(cmf) >= 0).all()
        t = np.arange(n, dtype=np.float64) / n * (1.0 - 2.0 ** -20)
        want = np.minimum(np.searchsorted(cmf.astype(np.float64), t, side="right"), n - 1)   # first k with cmf[k] > t
        np.testing.assert_array_equal(guide2[b:b + n], want)
        # every random number of a bucket: the bisection's bin is not in front of the guide's place, and one window of eight from the
        # entry before that place nearly always holds it
        u = rng.random(64).astype(np.float32)
        u = np.concatenate([u, cmf[rng.integers(0, n, 16)], np.nextafter(cmf[rng.integers(0, n, 16)], np.float32(0))]).astype(np.float32)
        u = u[u < 1.0]
        for x in u:
            k = reference_bisection(cmf, x)
            j = min(int(np.float32(x) * np.float32(n)), n - 1)
            assert guide2[b + j] <= k, (b, n, x, k, j, guide2[b + j])
            assert k == int((cmf <= x).sum()) or k == n - 1       # what the windows count
            checked += 1
    assert checked > 1000


def test_first_stage_guide_and_gamma_q_are_their_definitions(gpu, pkg, ob):
    scene, r, o, tup = _setup(pkg, ob)
    et, lt, q, gamma = tup
    gamma = np.asarray(gamma, np.float32).reshape(pkg.NUM_SUBSPACE, pkg.NUM_SUBSPACE)
    q = np.asarray(q, np.float32)
    r.launch("light trace", 7)
    r.build_sampler()
    _, guide1, gamma_q = r.sampling_tables(0)
    t = (np.arange(1024, dtype=np.float32) / np.float32(1024))
    rng = np.random.default_rng(6)
    for e in rng.integers(0, pkg.NUM_SUBSPACE, 40):
        row = gamma[e]
        np.testing.assert_array_equal(guide1[e], np.searchsorted(row, t, side="right"))
        for x in rng.random(32).astype(np.float32):
            assert guide1[e][int(x * np.float32(1024))] <= reference_bisection(row, x)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.concatenate([gamma[:, :1], gamma[:, 1:] - gamma[:, :-1]], axis=1) / q[None, :]
    fin = np.isfinite(g)
    np.testing.assert_array_equal(gamma_q[fin].view(np.uint32), g[fin].view(np.uint32))   # the same FP32 subtraction and division, bit for bit
    assert (np.isnan(g) == np.isnan(gamma_q)).all() and (np.isinf(g) == np.isinf(gamma_q)).all()


def test_radix_sort_build_writes_the_same_guide(gpu, pkg, ob):
    """SPCBPT_SAMPLER_BUILD=hipcub (the radix-sort form of LVC_Process) builds the guide table with a kernel of its own."""
    scene, r, o, tup = _setup(pkg, ob)
    r.launch("light trace", 9)
    lvc = r.lvc_read()
    r.lvc_import(lvc); r.build_sampler()
    sub, cmfs, jump, vc, pc = r.sampler_read()
    g_counting, _, _ = r.sampling_tables(vc)
    os.environ["SPCBPT_SAMPLER_BUILD"] = "hipcub"
    try:
        r2 = pkg.Renderer(scene, 0)
    finally:
        del os.environ["SPCBPT_SAMPLER_BUILD"]
    r2.resize(32, 24)
    r2.set_light_trace(4000, 64, 1)
    r2.set_subspace(*tup)
    r2.lvc_import(lvc); r2.build_sampler()
    sub2, cmfs2, jump2, vc2, pc2 = r2.sampler_read()
    assert vc2 == vc
    np.testing.assert_array_equal(jump2, jump)
    g_radix, _, _ = r2.sampling_tables(vc)
    same_cmf = np.array_equal(cmfs2, cmfs)
    if same_cmf:
        np.testing.assert_array_equal(g_radix, g_counting)
    else:   # (the two builds sum a subspace's weights differently: the CMFs may differ in the last place, and a guide entry with them)
        assert (g_radix.astype(np.int64) - g_counting.astype(np.int64)).__abs__().max() <= 1
    # ... and the frames rendered through it are the frames of the counting build
    for x in (r, r2):
        cam = scene.camera
        x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 4 / 3)
    if same_cmf:
        r.clear_accum(); r2.clear_accum()
        r.launch("SPCBPT_eye", 3); r2.launch("SPCBPT_eye", 3)
        np.testing.assert_array_equal(r.read_accum(), r2.read_accum())
