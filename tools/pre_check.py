"""Developer check (GPU): full preprocessing on a small budget, then SPCBPT with the trained tuple vs PT."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
p = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
scene = {"cornell": p.scenes.cornell_box, "bedroom": lambda: p.scenes.bedroom(200000, tex_size=256), "hallway": p.scenes.hallway}[name]()
W, H = 256, 256
r = p.Renderer(scene, 0)
cam = scene.camera
r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
r.resize(W, H)
r.set_light_trace(100000, 52, 1)
N = 64
for f in range(N): r.launch("pt", f)
pt = r.read_accum()[..., :3].astype(np.float64)
t = time.time()
target = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
r.preprocess(target, target, True)
print("preprocess s", time.time() - t)
et, lt, q, gm = r.get_subspace()
paths, nodes = r.train_records()
print("records", len(paths), len(nodes), "eye tree", len(et), "labels", len(set(et["label"][et["leaf"] == 1])), "light tree", len(lt),
      "labels", len(set(lt["label"][lt["leaf"] == 1])), "Q finite", int((q < 1e30).sum()))
r.clear_accum()
for f in range(N): r.render_frame("SPCBPT_eye", f, launch_frame=1000 + f)
sp = r.read_accum()[..., :3].astype(np.float64)
print("pt mean", pt.mean(), "spcbpt(trained) mean", sp.mean(), "ratio", sp.mean() / pt.mean())
r.set_subspace()
r.clear_accum()
for f in range(N): r.render_frame("SPCBPT_eye", f, launch_frame=1000 + f)
sm = r.read_accum()[..., :3].astype(np.float64)
ref = 0.5 * (pt + sm)
print("ratio minimal", sm.mean() / pt.mean())
def rm(a, b): return float(np.sqrt(((a - b) ** 2).mean()))
print("rmse pt vs minimal", rm(pt, sm), " pt vs trained", rm(pt, sp), " minimal vs trained", rm(sm, sp))
