"""developer: the interactive loop by itself on the bench scene (for a kernel trace: tools/viewer_timeline.sh) -- N viewer frames from a
steady view in the default pipeline mode, then the reference's order (mode 0)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scenes.bedroom(target_tris=1_000_000)
W, H = 1920, 1080
os.environ["SPCBPT_EYE_BATCH"] = "1"
r = pkg.Renderer(scene, 0)
cam = scene.camera
r.set_camera_lookat(cam["eye"], cam["lookat"], cam["up"], cam["fov"], W / H)
r.resize(W, H)
r.set_light_trace(100000, 52, 1)
r.preprocess(2_000_000, 2_000_000, True)
for mode in (2, 0):
    v = pkg.api.Viewer(r, cam["eye"], cam["lookat"], cam["up"], cam["fov"], W, H)
    v.set_pipeline(mode)
    for f in range(4): v.frame()
    t0 = time.perf_counter()
    n = 12
    for f in range(n): v.frame()
    print("mode", mode, "ms per displayed frame", round((time.perf_counter() - t0) / n * 1e3, 3))
    v.close()
r.sync()
