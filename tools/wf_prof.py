"""A few wavefront frames of the bench scene, for rocprofv3 --kernel-trace --stats (dev tool)."""
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
n = 4
for f in range(n):
    r.render_frame("SPCBPT_eye", f)
r.sync()
print("frames", n)
