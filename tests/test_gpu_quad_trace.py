"""The quad-lane traversal prototype (csrc/quad_trace.hip: four lanes per ray, one coalesced 64-B node fetch per visit) returns the
hits of the lane-per-ray loop and of the oracle's BVH: same triangle for >= 99.9 % of the rays, |dt| <= 1e-5 max(1, t) -- the bar of
tests/test_gpu_parity.py::test_traversal_matches_oracle_bvh -- and between the two device schedules the SAME triangle, distance and
barycentrics bit for bit wherever the ray is not within rounding of a tie (they share the slab arithmetic and the triangle test)."""
import numpy as np
import pytest

from tests.test_gpu_parity import _pair, _rays

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_name,kw", [("cornell_box", {}), ("simple_room", {"n": 6}), ("bedroom", {"target_tris": 60000, "tex_size": 64})])
def test_quad_traversal_matches_lane_traversal_and_oracle(gpu, pkg, ob, scene_name, kw):
    scene = getattr(pkg.scenes, scene_name)(**kw)
    r, o = _pair(pkg, ob, scene, 8, 8)
    lo, hi = scene.vertices.min(0) + 0.05, scene.vertices.max(0) - 0.05
    rays = _rays(np.random.default_rng(11), 50000, lo, hi)
    (t0, tri0, uv0), _, st0 = r.trace_bench(rays, 0, False, repeat=1)
    to, trio, uvo = o.trace_closest(rays)
    ts, tris, uvs = r.trace_closest(rays)                                   # the stand-alone one-ray-per-lane kernel
    assert np.array_equal(tri0, tris) and np.array_equal(t0, ts)            # the pool-fed lane kernel IS that loop
    rays2 = _rays(np.random.default_rng(12), 50000, lo, hi, tmax=np.random.default_rng(13).uniform(0.05, 4.0, 50000))
    v0, _, _ = r.trace_bench(rays2, 0, True, repeat=1)
    assert np.array_equal(v0, r.trace_any(rays2))
    vo = o.trace_any(rays2)
    for mode in (1, 2, 3, 4):                                               # one, two, four rays per quad in flight; 4 = the lean node step
        (t1, tri1, uv1), _, st1 = r.trace_bench(rays, mode, False, repeat=1)
        same = tri1 == tri0
        assert same.mean() >= 0.9995, (mode, same.mean())
        assert np.array_equal(t1[same], t0[same]) and np.array_equal(uv1[same], uv0[same]), mode
        assert (tri1 == trio).mean() >= 0.999, mode
        ok = tri1 == trio
        assert (np.abs(t1 - to)[ok] <= 1e-5 * np.maximum(1.0, to[ok])).all(), mode
        # the schedules visit the same nodes (ranking the four keys across the quad IS the sort of the lane kernel)
        if mode != 4:
            assert abs(st1["node_visits"] - st0["node_visits"]) <= 0.002 * st0["node_visits"], mode
        else:   # the lean form continues with the nearest child but pushes the other hits in slot order, not sorted: a few more visits
            assert st0["node_visits"] * 0.998 <= st1["node_visits"] <= 1.12 * st0["node_visits"], (st0["node_visits"], st1["node_visits"])
        assert st1["tri_tests"] <= st0["tri_tests"] * 1.6, mode             # a leaf is tested whole: a hit cannot skip its later triangles
        v1, _, _ = r.trace_bench(rays2, mode, True, repeat=1)
        assert (v1 == v0).mean() >= 0.9995, mode
        assert (v1 == vo).mean() >= 0.999, mode


def test_quad_traversal_edge_cases(gpu, pkg, ob):
    scene = pkg.scenes.cornell_box()
    r, o = _pair(pkg, ob, scene, 8, 8)
    # one ray; fewer rays than a wave holds quads; rays that leave the scene; ray counts that are not a multiple of anything
    up = np.array([[0, 1.0, 0, 1e-3, 0, 1, 0, 1e16]], np.float32)
    (t, tri, uv), _, _ = r.trace_bench(up, 1, False, repeat=1)
    assert tri[0] >= len(scene.indices) and abs(t[0] - 0.998) < 1e-4          # the light quad from its front
    out = np.array([[0, 1.0, 3.0, 1e-3, 0, 0, 1, 1e16]] * 7, np.float32)      # through the open front of the box
    (t, tri, uv), _, _ = r.trace_bench(out, 1, False, repeat=1)
    assert (tri == -1).all() and (t == np.float32(1e16)).all()
    lo, hi = scene.vertices.min(0) + 0.05, scene.vertices.max(0) - 0.05
    for n in (3, 17, 63, 65, 1001):
        rays = _rays(np.random.default_rng(n), n, lo, hi)
        (t0, tri0, _), _, _ = r.trace_bench(rays, 0, False, repeat=1)
        for mode in (1, 2, 3, 4):
            (t1, tri1, _), _, _ = r.trace_bench(rays, mode, False, repeat=1)
            assert np.array_equal(tri0, tri1) and np.array_equal(t0, t1), (n, mode)
    # single-sided emitters for path rays, opaque to shadow rays (q16), in the quad kernel too
    down = np.array([[0, 1.9995, 0, 1e-4, 0, -1, 0, 1e16]], np.float32)
    (t, tri, _), _, _ = r.trace_bench(down, 1, False, repeat=1)
    assert tri[0] < len(scene.indices)
    vis, _, _ = r.trace_bench(np.array([[0, 1.9995, 0, 1e-4, 0, -1, 0, 1.0]], np.float32), 1, True, repeat=1)
    assert vis[0] == 0


def test_trace_bench_rejects_bad_arguments(gpu, pkg):
    r = pkg.Renderer(pkg.scenes.cornell_box(), 0)
    rays = np.array([[0, 1.0, 0, 1e-3, 0, 1, 0, 1e16]], np.float32)
    with pytest.raises(pkg.SpcbptError):
        r.trace_bench(rays, 7, False, repeat=1)
    with pytest.raises(pkg.SpcbptError):
        r.trace_bench(rays, 1, False, repeat=0)
