"""Traversal in isolation on the bench scene: primary, diffuse-bounce and LVC shadow rays through k_trace_closest / k_trace_any.
Run under `rocprofv3 --kernel-trace --stats` to get kernel times; prints the ray counts so Grays/s can be derived."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
r.launch("light trace", 1); r.build_sampler()
lvc = r.lvc_read()
U, V, Wv = p.api.camera_frame(c["eye"], c["lookat"], c["up"], c["fov"], W / H); eye = np.asarray(c["eye"], np.float32)
rng = np.random.default_rng(1)
# primary rays in 8x8 tile order like the megakernel
ty, tx = np.meshgrid(np.arange(H // 8), np.arange(W // 8), indexing="ij")
py = (ty[..., None, None] * 8 + np.arange(8)[:, None]).repeat(8, -1).reshape(-1)
px = (tx[..., None, None] * 8 + np.arange(8)[None, :]).repeat(8, -2).reshape(-1)
d = ((px + 0.5) / W * 2 - 1)[:, None] * U + ((py + 0.5) / H * 2 - 1)[:, None] * V + Wv
d /= np.linalg.norm(d, axis=1, keepdims=True)
n = d.shape[0]
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = eye; rays[:, 3] = 1e-3; rays[:, 4:7] = d; rays[:, 7] = 1e16
t, tri, uv = r.trace_closest(rays)
hit = tri >= 0
print("primary rays", n, "hit fraction", hit.mean())
P = rays[:, 0:3] + t[:, None] * rays[:, 4:7]
vtx = np.asarray(scene.vertices, np.float32).reshape(-1, 3); idx = np.asarray(scene.indices).reshape(-1, 3)
tt = np.where(hit & (tri < idx.shape[0]), tri, 0)
N = np.cross(vtx[idx[tt, 1]] - vtx[idx[tt, 0]], vtx[idx[tt, 2]] - vtx[idx[tt, 0]])
N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)
N *= np.where((N * rays[:, 4:7]).sum(1, keepdims=True) > 0, -1, 1)
# cosine-distributed bounce
u1, u2 = rng.random(n), rng.random(n)
a = np.where(np.abs(N[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
T = np.cross(N, a); T /= np.linalg.norm(T, axis=1, keepdims=True); B = np.cross(N, T)
rr, ph = np.sqrt(u1), 2 * np.pi * u2
d2 = (rr * np.cos(ph))[:, None] * T + (rr * np.sin(ph))[:, None] * B + np.sqrt(1 - u1)[:, None] * N
sec = np.zeros((n, 8), np.float32); sec[:, 0:3] = P + 1e-3 * N; sec[:, 3] = 1e-3; sec[:, 4:7] = d2; sec[:, 7] = 1e16
sec = sec[hit]
t2, tri2, _ = r.trace_closest(sec)
print("secondary rays", sec.shape[0], "hit fraction", (tri2 >= 0).mean())
# shadow rays to random LVC vertices
k = rng.integers(0, lvc.shape[0], n)
L = lvc["position"][k]
dv = L - P; dist = np.linalg.norm(dv, axis=1)
sh = np.zeros((n, 8), np.float32); sh[:, 0:3] = P + 1e-3 * N; sh[:, 3] = 1e-3; sh[:, 4:7] = dv / dist[:, None]; sh[:, 7] = dist - 2e-3
sh = sh[hit]
for rep in range(3):
    vis = r.trace_any(sh)
print("shadow rays", sh.shape[0], "visible fraction", vis.mean())
for rep in range(2):
    r.trace_closest(sec); r.trace_closest(rays)
