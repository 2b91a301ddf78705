"""Developer check for tests/test_gpu_configs.py (C5): batch means of PT and SPCBPT on the reduced hallway with a trained tuple, to
size the sample counts and the tolerance of the unbiasedness test (batch k = frames [k n/K, (k+1) n/K), recovered from the running
mean at checkpoints).  Usage: python tools/hallway_means.py [tris] [pt_frames] [sp_frames] [uniform]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
p = g.load_package()
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
NPT = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
NSP = int(sys.argv[3]) if len(sys.argv) > 3 else 600
scene = p.scenes.hallway(target_tris=tris)
W, H, K = 256, 144, 8
r = p.Renderer(scene, 0)
c = scene.camera
r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H); r.resize(W, H)
r.set_light_trace(20000, 52, 1)
r.set_pretrace(20000, 10)
r.preprocess(target_paths=200000, target_q_paths=200000, train=True)

def batch_means(step, n):
    r.clear_accum()
    cum, out, prev_m, prev_n = [], [], 0.0, 0
    for k in range(K):
        hi = (k + 1) * n // K
        for f in range(prev_n if k else 0, hi): step(f)
        m = r.read_accum()[..., :3].astype(np.float64).mean()
        out.append((hi * m - prev_n * prev_m) / (hi - prev_n)); prev_m, prev_n = m, hi
    return np.array(out), prev_m

def report(name, b, m):
    print(f"{name}: mean {m:.6g}  batch means {np.array2string(b, precision=6)}  std of a batch {b.std(ddof=1):.3g} "
          f"-> std error of the mean {b.std(ddof=1) / np.sqrt(K):.3g} ({b.std(ddof=1) / np.sqrt(K) / m * 100:.2f} %)")

b, m_pt = batch_means(lambda f: r.launch("pt", f), NPT); report("pt", b, m_pt)
b, m_sp = batch_means(lambda f: r.render_frame("SPCBPT_eye", f, launch_frame=100000 + f), NSP); report("spcbpt trained", b, m_sp)
print("rel diff %.4f" % (abs(m_sp - m_pt) / m_pt))
if len(sys.argv) > 4:
    r.set_connection_sampler(1)
    b, m_u = batch_means(lambda f: r.render_frame("SPCBPT_eye", f, launch_frame=100000 + f), NSP); report("spcbpt uniformSample", b, m_u)
