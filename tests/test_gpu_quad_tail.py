"""The quad tail of the pooled traversal pass (csrc/device_lib.h, trace_pool): when the wave's ray pool is dry and at most 16 rays
are still in flight, each of them continues on FOUR lanes -- lane r loads record r of the node (one coalesced 64-B line per ray),
tests its child, the four entry distances are ranked across the quad with DPP rotations and pushed onto the owner lane's own
stack in push_far's order; at a leaf lane r tests triangle r.  Same keys, same order, same stack: the ray visits what it would have
visited and every hit is the lane loop's hit, so the FILM must be the lane loop's film bit for bit.  SPCBPT_NO_QUAD_TAIL=1 at
spcbpt_create switches the tail off (the lane loop runs to the end, as in rounds 1-3)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(pkg, scene, W, H, lt, tup, no_tail, frames=3, batch=False, counters=False, no_fan=False):
    if no_tail: os.environ["SPCBPT_NO_QUAD_TAIL"] = "1"
    if no_fan: os.environ["SPCBPT_NO_FAN_TAIL"] = "1"
    if batch: os.environ["SPCBPT_EYE_BATCH"] = "4"
    try:
        r = pkg.Renderer(scene, 0)
    finally:
        os.environ.pop("SPCBPT_NO_QUAD_TAIL", None); os.environ.pop("SPCBPT_NO_FAN_TAIL", None); os.environ.pop("SPCBPT_EYE_BATCH", None)
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
    if counters:
        r.enable_counters(2); r.reset_counters()
    if batch:
        r.set_light_ahead(True)
        for f in range(frames):
            r.launch("light trace", f + 1); r.build_sampler()
        r.launch_eye_batch(list(range(frames)))
    else:
        for f in range(frames):
            r.render_frame("SPCBPT_eye", f)
    r.sync()
    return r, r.read_accum().copy(), tup


@pytest.mark.parametrize("name", ["cornell", "bedroom", "hallway", "needles"])
def test_quad_tail_leaves_every_film_bit_identical(gpu, pkg, name):
    scene, W, H, lt = {"cornell": (pkg.scenes.cornell_box(), 192, 192, (4000, 64, 1)),
                       "bedroom": (pkg.scenes.bedroom(target_tris=60000, tex_size=64), 256, 144, (8000, 64, 1)),
                       "hallway": (pkg.scenes.hallway(target_tris=20000), 256, 144, (8000, 52, 1)),
                       # slivers spanning the room: stacks run past their 16 LDS entries, so the tail pushes and pops the HBM part too
                       "needles": (pkg.scenes.needle_room(20000), 160, 120, (4000, 64, 1))}[name]
    _, off, tup = _render(pkg, scene, W, H, lt, None, True)
    r_on, on, _ = _render(pkg, scene, W, H, lt, tup, False, no_fan=True)    # one quad per ray to the end
    assert np.isfinite(on).all() and np.array_equal(on, off), int((on != off).any(-1).sum())
    _, fan, _ = _render(pkg, scene, W, H, lt, tup, False)                   # the default: shadow rays fan out over the idle quads
    assert np.array_equal(fan, off), int((fan != off).any(-1).sum())
    _, b_on, _ = _render(pkg, scene, W, H, lt, tup, False, batch=True)
    assert np.array_equal(b_on, off)                                      # ... and in the batched kernel


def test_quad_tail_is_in_use_and_counts_the_same_events(gpu, pkg):
    """The counting instantiation runs the tail too: node visits, triangle tests, rays and connections per frame are those of the lane
    loop exactly (the same traversal), while the lanes busy per node-step slot go up -- four per ray in the tail."""
    scene = pkg.scenes.bedroom(target_tris=60000, tex_size=64)
    lt = (8000, 64, 1)
    r_off, off, tup = _render(pkg, scene, 256, 144, lt, None, True, frames=2, counters=True)
    r_on, on, _ = _render(pkg, scene, 256, 144, lt, tup, False, frames=2, counters=True, no_fan=True)
    assert np.array_equal(on, off)
    c_off, c_on = r_off.counters(), r_on.counters()
    print({k: (c_on[k], c_off[k]) for k in c_on})
    for k in ("closest_rays", "shadow_rays", "node_visits", "surface_vertices", "connections", "tree_nodes", "cmf_probes"):
        assert c_on[k] == c_off[k], (k, c_on[k], c_off[k])
    # a leaf is tested whole by its quad, where the lane loop's any-hit ray stops at the first triangle it hits
    assert c_off["tri_tests"] <= c_on["tri_tests"] <= 1.05 * c_off["tri_tests"], (c_on["tri_tests"], c_off["tri_tests"])
    p_off, p_on = r_off.phase_clocks(), r_on.phase_clocks()
    assert p_on["node_lanes"] > 1.03 * p_off["node_lanes"]               # the tail's quads: four lanes per node visit
    assert p_on["node_slots"] <= p_off["node_slots"]


def test_fan_tail_is_in_use_and_changes_only_the_order_of_visits(gpu, pkg):
    """fan_tail (device_lib.h): a shadow ray's stack is a bag, so its nodes may be visited several at a time.  Rays, vertices and
    connections are those of the ordered traversal exactly and so is the film; an unoccluded ray visits the same nodes, an occluded
    one finds its occluder a few nodes earlier or later; the node-step iterations of the waves go DOWN (the point of it)."""
    scene = pkg.scenes.bedroom(target_tris=60000, tex_size=64)
    lt = (8000, 64, 1)
    r_q, q, tup = _render(pkg, scene, 256, 144, lt, None, False, frames=2, counters=True, no_fan=True)
    r_f, f, _ = _render(pkg, scene, 256, 144, lt, tup, False, frames=2, counters=True)
    assert np.array_equal(f, q)
    c_q, c_f = r_q.counters(), r_f.counters()
    print({k: (c_f[k], c_q[k]) for k in c_f})
    for k in ("closest_rays", "shadow_rays", "surface_vertices", "connections", "tree_nodes", "cmf_probes"):
        assert c_f[k] == c_q[k], (k, c_f[k], c_q[k])
    assert 0.97 * c_q["node_visits"] <= c_f["node_visits"] <= 1.05 * c_q["node_visits"], (c_f["node_visits"], c_q["node_visits"])
    assert c_f["tri_tests"] <= 1.08 * c_q["tri_tests"], (c_f["tri_tests"], c_q["tri_tests"])
    p_q, p_f = r_q.phase_clocks(), r_f.phase_clocks()
    print("node-step iterations", p_f["node_slots"] // 64, p_q["node_slots"] // 64)
    assert p_f["node_slots"] < 0.97 * p_q["node_slots"]
