"""Fan pairs (round 6; csrc/lbvh.cpp make_fan_pairs, csrc/dev_traversal.h: the triangle step of trace_pool and of traverse<>).
Inside every leaf the halves of a quad are made neighbours and the first half's 64-B pair slot holds (A.P0, A.P1, A.P2, B.P2): one
triangle step then tests A and B = (A.P0, A.P2, B.P2) -- the operations and the order of two steps.  SPCBPT_NO_TRI_PAIRS=1 at
spcbpt_create clears the pair flag of every slot (same triangle order, same device code, one test per step): films, standalone hits
and the event counts must be the same bit for bit / count for count, for SPCBPT (the pooled pass) and for "pt" (traverse<>)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(pkg, scene, W, H, lt, tup, pairs):
    if not pairs: os.environ["SPCBPT_NO_TRI_PAIRS"] = "1"
    try:
        r = pkg.Renderer(scene, 0)
    finally:
        os.environ.pop("SPCBPT_NO_TRI_PAIRS", None)
    cam = scene.camera
    r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
    r.resize(W, H)
    r.set_light_trace(*lt)
    if tup is None:
        r.set_pretrace(20000, 10)
        r.preprocess(target_paths=100000, target_q_paths=100000, train=True)
        tup = r.get_subspace()
    else:
        r.set_subspace(*tup)
    r.enable_counters(True); r.reset_counters()
    for f in range(3):
        r.render_frame("SPCBPT_eye", f)
    r.sync()
    sp, c_sp = r.read_accum().copy(), r.counters()
    r.clear_accum(); r.reset_counters()
    for f in range(2):
        r.render_frame("pt", f)
    r.sync()
    pt, c_pt = r.read_accum().copy(), r.counters()
    rng = np.random.default_rng(9)
    lo, hi = np.asarray(scene.vertices).min(0), np.asarray(scene.vertices).max(0)
    n = 6000
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([o, np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], axis=1)
    t, tri, uv = r.trace_closest(rays)
    short = rays.copy(); short[:, 7] = rng.uniform(0.05, 3.0, n).astype(np.float32)
    vis = r.trace_any(short)
    return dict(sp=sp, pt=pt, t=t.copy(), tri=tri.copy(), uv=uv.copy(), vis=vis.copy(), c_sp=c_sp, c_pt=c_pt, tup=tup)


@pytest.mark.parametrize("name", ["bedroom", "cornell", "needles"])
def test_pair_step_changes_no_film_no_hit_and_no_count(gpu, pkg, name):
    scene, W, H, lt = {"bedroom": (pkg.scenes.bedroom(target_tris=60000, tex_size=64), 256, 144, (8000, 64, 1)),   # grids of quads: nearly every triangle is half of a pair
                       "cornell": (pkg.scenes.cornell_box(), 128, 128, (3000, 64, 1)),                              # a dozen quads + the quad light's two emitter triangles (culling flags of A and B)
                       "needles": (pkg.scenes.needle_room(20000), 160, 120, (4000, 64, 1))}[name]                    # thin single triangles: hardly any pair
    a = _run(pkg, scene, W, H, lt, None, pairs=True)
    b = _run(pkg, scene, W, H, lt, a["tup"], pairs=False)
    assert np.isfinite(a["sp"]).all() and a["sp"][..., :3].mean() > 0
    assert np.array_equal(a["sp"], b["sp"])          # the pooled pass
    assert np.array_equal(a["pt"], b["pt"])          # traverse<>
    assert np.array_equal(a["tri"], b["tri"]) and np.array_equal(a["t"], b["t"]) and np.array_equal(a["uv"], b["uv"])
    assert np.array_equal(a["vis"], b["vis"])
    for k in ("closest_rays", "shadow_rays", "node_visits", "tri_tests", "surface_vertices", "connections"):
        assert a["c_pt"][k] == b["c_pt"][k], (k, a["c_pt"][k], b["c_pt"][k])            # one ray per lane to its end: deterministic visit for visit
        # (the pooled pass's fan-out tail visits an OCCLUDED ray's nodes in an order that depends on timing -- same answers, a few visits
        # more or less from run to run: its node and triangle counts are compared within 0.5 %)
        tol = 0.005 * b["c_sp"][k] if k in ("node_visits", "tri_tests") else 0
        assert abs(a["c_sp"][k] - b["c_sp"][k]) <= tol, (k, a["c_sp"][k], b["c_sp"][k])
