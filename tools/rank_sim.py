"""What one rank of an N-GPU job does per frame, on one GPU and without the exchange: 1/N of the light cores, every N-th 8-row
band.  Shows how the step time of a rank scales with N and with the number of frames in flight (SPCBPT_RENDER_STREAMS)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(sys.argv[1]); streams = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
os.environ["SPCBPT_RENDER_STREAMS"] = str(streams)
import __graft_entry__ as g
p = g.load_package()
scene = p.scenes.bedroom()
W, H, M = 1920, 1080, 100000
r = p.Renderer(scene, 0)
c = scene.camera
r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H)
r.resize(W, H)
r.set_light_trace(M, 52, 1)
r.set_subspace()
r.set_light_trace(M, 52, 1, core_begin=0, core_count=M // N)
rows = (0, H, N)
def step(f):
    r.launch("light trace", f + 1); r.build_sampler(); r.launch("SPCBPT_eye", f, rows)
for f in range(4): step(f)
r.sync()
t0 = time.perf_counter()
for f in range(steps): step(f)
r.sync()
dt = (time.perf_counter() - t0) / steps
print(f"N={N} streams={streams}: {dt * 1e3:.3f} ms per rank-frame  -> {N}-GPU job at {(W * H + M) / dt / 1e6:.1f} Mpaths/s if the exchange were free")
