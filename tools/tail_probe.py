import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
r.preprocess(target_paths=2_000_000, target_q_paths=2_000_000, train=True)
r.launch("light trace", 1); r.build_sampler()
r.enable_counters(2); r.reset_counters()
r.launch("SPCBPT_eye", 0); r.sync()
ph = r.phase_clocks()
c = r.counters()
print(ph)
tot = ph["regen"] + ph["closest"] + ph["shade"] + ph["shadow_pool"] + ph["connect"]
print({k: round(ph[k] / tot, 4) for k in ("regen", "closest", "shade", "shadow_pool", "connect")})
print("node util", ph["node_lanes"] / ph["node_slots"], "tri util", ph["tri_lanes"] / ph["tri_slots"])
print("tail share of node slots", ph["tail_slots"] / ph["node_slots"], "tail util", (ph["tail_closest_lanes"] + ph["tail_shadow_lanes"]) / max(ph["tail_slots"], 1),
      "closest share of tail lanes", ph["tail_closest_lanes"] / max(ph["tail_closest_lanes"] + ph["tail_shadow_lanes"], 1))
print("node iterations per frame", ph["node_slots"] / 64, "tri iterations", ph["tri_slots"] / 64)
print("connect phase: jobs per round of 64", ph["job_lanes"] / max(ph["job_slots"] / 64, 1), "rounds per frame", ph["job_slots"] / 64, "jobs per eye path", ph["job_lanes"] / (W * H))
