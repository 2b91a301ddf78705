"""The hottest BVH nodes in LDS (round 5; csrc/lbvh.cpp hot_nodes_first, csrc/kernels.hip k_spcbpt, csrc/device_lib.h trace_pool).
The builder numbers the 64 nodes of largest surface area 0 .. 63, the eye megakernel copies node records [0, 19) into LDS at block
start and the pooled pass reads those records from there (one FLAT request that goes to LDS or to memory per lane).  Renumbering is
a pure permutation of the node array and an LDS copy is a copy: neither may change which nodes a ray visits, in which order, or what
it hits.  SPCBPT_BVH_HOT_NODES=0 at spcbpt_create leaves the nodes in depth-first order (the table then holds the first 19 nodes of
that order -- a different set, equally valid): the films must be the same bit for bit, for SPCBPT (the pooled pass) and for "pt"
(traverse<>, which reads every node from memory), and so must the standalone traversal's hits."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _films(pkg, scene, W, H, lt, tup, hot):
    if hot is not None: os.environ["SPCBPT_BVH_HOT_NODES"] = str(hot)
    try:
        r = pkg.Renderer(scene, 0)
    finally:
        os.environ.pop("SPCBPT_BVH_HOT_NODES", None)
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
    for f in range(3):
        r.render_frame("SPCBPT_eye", f)
    r.sync()
    sp = r.read_accum().copy()
    r.clear_accum()
    for f in range(2):
        r.render_frame("pt", f)
    r.sync()
    pt = r.read_accum().copy()
    rng = np.random.default_rng(5)
    lo, hi = np.asarray(scene.vertices).min(0), np.asarray(scene.vertices).max(0)
    n = 4000
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([o, np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], axis=1)
    t, tri, uv = r.trace_closest(rays)
    return sp, pt, (t.copy(), tri.copy(), uv.copy()), tup, r.scene_info()


@pytest.mark.parametrize("name", ["bedroom", "needles"])
def test_renumbering_and_the_lds_copy_change_no_film(gpu, pkg, name):
    scene, W, H, lt = {"bedroom": (pkg.scenes.bedroom(target_tris=60000, tex_size=64), 256, 144, (8000, 64, 1)),
                       "needles": (pkg.scenes.needle_room(20000), 160, 120, (4000, 64, 1))}[name]
    sp0, pt0, hits0, tup, info0 = _films(pkg, scene, W, H, lt, None, 0)       # depth-first numbering throughout
    sp1, pt1, hits1, _, info1 = _films(pkg, scene, W, H, lt, tup, None)       # the default: the 64 largest nodes first
    sp2, pt2, hits2, _, _ = _films(pkg, scene, W, H, lt, tup, 7)              # a crown smaller than the kernel's table
    assert info0 == info1
    assert np.isfinite(sp1).all() and sp1[..., :3].mean() > 0
    for sp, pt, hits in ((sp1, pt1, hits1), (sp2, pt2, hits2)):
        assert np.array_equal(sp, sp0), int((sp != sp0).any(-1).sum())
        assert np.array_equal(pt, pt0), int((pt != pt0).any(-1).sum())
        assert all(np.array_equal(a, b) for a, b in zip(hits, hits0))
