import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
p = g.load_package()
from oracle import binding as ob
from tests.parity_util import image_parity
scene = p.scenes.cornell_box()
W = H = 128
r = p.Renderer(scene, 0); o = ob.Oracle(scene)
cam = scene.camera
for x in (r, o):
    x.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], 1.0); x.resize(W, H); x.set_light_trace(20000, 52, 1)
r.set_pretrace(20000, 10)
r.preprocess(40000, 40000, True)
et, lt, q, cmf = r.get_subspace()
o.set_subspace(et, lt, q, cmf); o.set_cmf_double(True)
r.launch("light trace", 1); o.launch("light trace", 1)
a, b = r.lvc_read(), o.lvc_read()
print("lvc", len(a), len(b))
n = min(len(a), len(b))
same = (a["path_id"][:n] == b["path_id"][:n]) & (a["depth"][:n] == b["depth"][:n])
print("aligned frac", same.mean(), "first div", int(np.argmin(same)) if not same.all() else n)
m = same
print("subspace eq", (a["subspace_id"][:n][m] == b["subspace_id"][:n][m]).mean(), "lastzone eq", (a["last_zone_id"][:n][m] == b["last_zone_id"][:n][m]).mean())
rel = np.abs(a["rmis_pointer"][:n][m] - b["rmis_pointer"][:n][m]) / (np.abs(b["rmis_pointer"][:n][m]) + 1e-20)
print("rmis_pointer rel p50 p90 p99", np.percentile(rel, [50, 90, 99]))
rel = np.abs(a["pdf"][:n][m] - b["pdf"][:n][m]) / (np.abs(b["pdf"][:n][m]) + 1e-30)
print("pdf rel p99", np.percentile(rel, 99))
# identical LVC on both: import oracle's
r.lvc_import(b); r.build_sampler(); o.build_sampler()
sg, so = r.sampler_read(), o.sampler_read()
print("sampler eq", (sg[2] == so[2]).all(), np.abs(sg[1] - so[1]).max())
r.launch("SPCBPT_eye", 0); o.launch("SPCBPT_eye", 0)
s = image_parity(r.read_accum()[..., :3], o.read_accum()[..., :3])
print("same-LVC parity", s)
x, y = r.read_accum()[..., :3].astype(np.float64), o.read_accum()[..., :3].astype(np.float64)
d = np.abs(x - y).sum(-1) / (np.abs(y).sum(-1) + 1e-9)
print("rel diff percentiles", np.percentile(d, [50, 80, 90, 95, 99]))
