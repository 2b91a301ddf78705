"""Wavefront vs megakernel eye pass: bit equality of the accum buffer and timing (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
p = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "bedroom"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
scene = {"cornell": p.scenes.cornell_box, "bedroom": p.scenes.bedroom, "hallway": p.scenes.hallway}[name]()
out = {}
for mode in ("megakernel", "wavefront"):
    os.environ["SPCBPT_EYE_PASS"] = mode
    r = p.Renderer(scene, 0)
    c = scene.camera
    r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H)
    r.resize(W, H)
    r.set_light_trace(100000, 52, 1)
    r.set_subspace()
    if "--trained" in sys.argv: r.preprocess(400000, 400000, True)
    r.enable_counters(True); r.reset_counters()
    r.render_frame("SPCBPT_eye", 0)
    r.sync()
    cnt = r.counters()
    r.enable_counters(False)
    r.clear_accum()
    r.enable_kernel_timing(True)
    for f in range(6):
        if f == 2: r.reset_kernel_time()
        r.render_frame("SPCBPT_eye", f)
    r.sync()
    t = r.kernel_time("spcbpt_render")
    out[mode] = (r.read_accum().copy(), cnt, t)
    print(mode, "ms/frame", t, flush=True)
    del r
a, b = out["megakernel"][0], out["wavefront"][0]
print("bit-identical:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), "max abs diff", float(np.abs(a - b).max()), "mismatching pixels", int((a != b).any(-1).sum()))
print("counters mega", out["megakernel"][1]); print("counters wave", out["wavefront"][1])
