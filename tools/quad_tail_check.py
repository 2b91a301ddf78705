"""The quad tail of the pooled traversal pass (device_lib.h, trace_pool) must not change a single hit: films rendered with it and
without it (SPCBPT_NO_QUAD_TAIL at spcbpt_create) are compared bit for bit, single-frame and batched launches, two scenes."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()


def render(scene, W, H, lt, tup, no_tail, frames=4, batch=False):
    if no_tail: os.environ["SPCBPT_NO_QUAD_TAIL"] = "1"
    if batch: os.environ["SPCBPT_EYE_BATCH"] = "4"
    try:
        r = pkg.Renderer(scene, 0)
    finally:
        os.environ.pop("SPCBPT_NO_QUAD_TAIL", None); os.environ.pop("SPCBPT_EYE_BATCH", None)
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
    if batch:
        r.set_light_ahead(True)
        for f in range(frames):
            r.launch("light trace", f + 1); r.build_sampler()
        r.launch_eye_batch(list(range(frames)))
    else:
        for f in range(frames):
            r.render_frame("SPCBPT_eye", f)
    r.sync()
    return r.read_accum().copy(), tup


for name, scene, W, H, lt in (("cornell", pkg.scenes.cornell_box(), 256, 256, (4000, 64, 1)),
                              ("bedroom 60k", pkg.scenes.bedroom(target_tris=60000, tex_size=64), 320, 180, (8000, 64, 1)),
                              ("hallway 20k", pkg.scenes.hallway(target_tris=20000), 256, 144, (8000, 52, 1))):
    a, tup = render(scene, W, H, lt, None, True)
    b, _ = render(scene, W, H, lt, tup, False)
    c, _ = render(scene, W, H, lt, tup, False, batch=True)
    d, _ = render(scene, W, H, lt, tup, True, batch=True)
    print(name, "single: identical" if np.array_equal(a, b) else f"single: DIFFERENT in {(a != b).any(-1).sum()} pixels",
          "| batched: identical" if np.array_equal(c, d) else f"| batched: DIFFERENT in {(c != d).any(-1).sum()} pixels",
          "| batched == single" if np.array_equal(a, d) else "| batched != single", "mean", float(a[..., :3].mean()))
