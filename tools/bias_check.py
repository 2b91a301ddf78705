"""Developer experiment: SPCBPT-vs-PT mean image for several light-pass sizes (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
p = g.load_package()
scene = p.scenes.cornell_box()
W = H = 256
r = p.Renderer(scene, 0)
cam = scene.camera
r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0)
r.resize(W, H)
N = 128
for f in range(N): r.launch("pt", f)
pt = r.read_accum()[..., :3].astype(np.float64)
print("pt mean", pt.mean(axis=(0, 1)), pt.mean())
for M in (2000, 20000, 100000, 400000):
    r.set_light_trace(M, 52, 1)
    r.set_subspace()
    r.clear_accum()
    for f in range(N): r.render_frame("SPCBPT_eye", f)
    sp = r.read_accum()[..., :3].astype(np.float64)
    sub, cmfs, jump, vc, pc = r.sampler_read()
    print("M", M, "spcbpt mean", sp.mean(axis=(0, 1)), "ratio", sp.mean() / pt.mean(), "verts", vc, "max subspace", sub["size"].max())
