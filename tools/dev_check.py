"""Developer check on a GPU box: product vs oracle on small cases (not a test; see tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
p = g.load_package()
from oracle import binding as ob
from tests.parity_util import minimal_tuple, image_parity, rmse

def setup(scene, W, H, lt=(2000, 64, 1)):
    r = p.Renderer(scene, 0)
    o = ob.Oracle(scene)
    cam = scene.camera
    for x in (r, o):
        x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
        x.resize(W, H)
        x.set_light_trace(*lt)
    return r, o

scene = p.scenes.cornell_box() if len(sys.argv) < 2 or sys.argv[1] == "cornell" else p.scenes.simple_room()
W = H = 64
r, o = setup(scene, W, H)
print("scene info", r.scene_info())
# rays
rng = np.random.default_rng(1)
n = 20000
org = rng.uniform([-0.9, 0.1, -0.9], [0.9, 1.9, 0.9], size=(n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1)
t0, tri0, uv0 = o.trace_closest(rays)
t1, tri1, uv1 = r.trace_closest(rays)
print("closest: tri mismatch", int((tri0 != tri1).sum()), "max |dt|", float(np.abs(t0 - t1)[tri0 == tri1].max()))
rays2 = rays.copy(); rays2[:, 7] = rng.uniform(0.2, 3.0, size=n)
v0 = o.trace_any(rays2); v1 = r.trace_any(rays2)
print("any: mismatch", int((v0 != v1).sum()))
# PT parity
for f in range(4):
    r.launch("pt", f); o.launch("pt", f)
a, b = r.read_accum()[..., :3], o.read_accum()[..., :3]
print("pt parity", image_parity(a, b))
# light trace parity
tup = minimal_tuple(o, 2)
r.set_subspace(*tup); o.set_subspace(*tup)
r.launch("light trace", 1); o.launch("light trace", 1)
lg, lo = r.lvc_read(), o.lvc_read()
print("lvc counts", len(lg), len(lo))
if len(lg) == len(lo):
    same = (lg["path_id"] == lo["path_id"]) & (lg["depth"] == lo["depth"])
    print("lvc same order", same.mean())
    for k in ("position", "normal", "flux", "pdf", "single_pdf", "rmis_pointer", "last_lum", "color"):
        x, y = lg[k].astype(np.float64), lo[k].astype(np.float64)
        rel = np.abs(x - y) / (np.abs(y) + 1e-12)
        print("  ", k, "max rel", float(rel.max()), "p99", float(np.percentile(rel, 99)))
    print("   subspace eq", (lg["subspace_id"] == lo["subspace_id"]).mean(), "mat eq", (lg["material_id"] == lo["material_id"]).mean())
# sampler parity on identical LVC
r.lvc_import(lo); r.build_sampler(); o.build_sampler()
sg, so = r.sampler_read(), o.sampler_read()
print("sampler: vc/pc", sg[3:], so[3:], "sizes eq", (sg[0]["size"] == so[0]["size"]).all(), "bias eq", (sg[0]["jump_bias"] == so[0]["jump_bias"]).all(),
      "jump eq", (sg[2] == so[2]).all(), "cmf max abs", float(np.abs(sg[1] - so[1]).max()))
# SPCBPT parity
for mode in (0, 1):
  o.set_cmf_double(mode)
  r.clear_accum(); o.clear_accum()
  for f in range(4):
      r.render_frame("SPCBPT_eye", f); o.render_frame("SPCBPT_eye", f)
  a, b = r.read_accum()[..., :3], o.read_accum()[..., :3]
  print("spcbpt parity cmf_double=%d" % mode, image_parity(a, b))
r.enable_counters(True); r.reset_counters(); o.reset_counters()
r.render_frame("SPCBPT_eye", 5); o.render_frame("SPCBPT_eye", 5)
cg, co = r.counters(), o.counters()
for k in cg: print("  ", k, cg[k], co[k])
