"""Developer check at the bench scale (bedroom, 1920x1080, trained tuple): SPCBPT vs PT means, run-to-run reproducibility of the
frame-by-frame loop, and equality of batched and frame-by-frame films."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPCBPT_EYE_BATCH"] = "4"
import numpy as np
import __graft_entry__ as g
p = g.load_package()
scene = p.scenes.bedroom()
W, H = 1920, 1080
r = p.Renderer(scene, 0)
c = scene.camera
r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H); r.resize(W, H)
r.set_light_trace(100000, 52, 1)
r.preprocess(2000000, 2000000, True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def plain():
    r.clear_accum()
    for f in range(N): r.render_frame("SPCBPT_eye", f)
    r.sync(); return r.read_accum()[..., :3].copy()
def batched(F):
    r.clear_accum(); q = []
    for f in range(N):
        r.launch("light trace", f + 1); r.build_sampler(); q.append(f)
        if len(q) == F or f == N - 1: r.launch_eye_batch(q); q = []
    r.sync(); return r.read_accum()[..., :3].copy()
a1, a2 = plain(), plain()
b4, b1 = batched(4), batched(1)
def diff(x, y):
    d = np.abs(x.astype(np.float64) - y); return f"equal={np.array_equal(x, y)} differing pixels={int((d.max(axis=2) > 0).sum())} max abs={d.max():.3g}"
print("plain vs plain   ", diff(a1, a2))
print("batch4 vs plain  ", diff(b4, a1))
print("batch1 vs plain  ", diff(b1, a1))
print("batch4 vs batch4 ", diff(b4, batched(4)))
for f in range(N): r.launch("pt", f)
