"""Times the kernels next to the eye megakernel on the bench scene: the light pass (one launch, synchronised) and a path-tracing
frame ("pt"), which share the one-ray-per-lane traversal loop of device_lib.h (traverse<>).  usage: [SPCBPT_LIB=...] python tools/aux_kernels_time.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scenes.bedroom(target_tris=1_000_000)
r = pkg.Renderer(scene, 0)
cam = scene.camera
W, H = 1920, 1080
r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
r.resize(W, H)
r.set_light_trace(100000, 52, 1)
r.set_subspace()
def timed(fn, n):
    fn(); r.sync()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    r.sync()
    return (time.perf_counter() - t0) / n * 1e3
k = [0]
def light(i=0):
    k[0] += 1; r.launch("light trace", k[0]); r.sync()
def pt(i=0):
    k[0] += 1; r.launch("pt", k[0])
print("light pass ms (launch + sync)", round(timed(light, 50), 4), "pt frame ms", round(timed(pt, 30), 4))
r.sync()
print("pt checksum", float(np.float64(r.read_accum()).sum()))
