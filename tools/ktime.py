"""HIP-event time of the SPCBPT_eye megakernel on the bench scene, one frame in flight: plain build, then counting build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
p = g.load_package()
scene = p.scenes.bedroom()
W, H = 1920, 1080
r = p.Renderer(scene, 0)
c = scene.camera
r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H)
r.resize(W, H)
r.set_light_trace(100000, 52, 1)
r.set_subspace()
if "--trained" in sys.argv: r.preprocess(2000000, 2000000, True)
r.render_frame("SPCBPT_eye", 0)
r.sync()
for counting in (False, True):
    r.enable_counters(counting)
    r.enable_kernel_timing(True); r.reset_kernel_time()
    n = 6
    for f in range(1, 1 + n):
        r.launch("SPCBPT_eye", f)
        r.sync()
    t = r.kernel_time("spcbpt_render")
    print("counting" if counting else "plain", t)
