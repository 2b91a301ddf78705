"""Share of wave-clocks the SPCBPT_eye megakernel spends per phase (counting build), bench scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
if "--trained" in sys.argv: r.preprocess(2000000, 2000000, True)
r.render_frame("SPCBPT_eye", 0)
r.enable_counters(2 if "--executed" in sys.argv else True); r.reset_counters()
r.launch("SPCBPT_eye", 1)
r.sync()
ph = r.phase_clocks()
names = ("regen", "closest", "shade", "shadow_pool", "connect")
tot = sum(ph[k] for k in names)
print({k: round(ph[k] / tot, 3) for k in names})
cnt = r.counters()
print("resampling lane-clocks per vertex", ph["sample_lane_clocks"] / max(1, cnt["surface_vertices"]), "shade wave-clocks per wave-vertex (64 lanes)", ph["shade"] / max(1, cnt["surface_vertices"] / 64))
print("node-loop lane utilisation", round(ph["node_lanes"] / max(1, ph["node_slots"]), 3), "tri-loop", round(ph["tri_lanes"] / max(1, ph["tri_slots"]), 3),
      "wave node iterations", ph["node_slots"] // 64, "wave tri iterations", ph["tri_slots"] // 64)
print("pooled pass after its pool ran dry: share of the node-step slots", round(ph["tail_slots"] / max(1, ph["node_slots"]), 3),
      "lanes active there: closest", round(ph["tail_closest_lanes"] / max(1, ph["tail_slots"]), 3), "shadow", round(ph["tail_shadow_lanes"] / max(1, ph["tail_slots"]), 3),
      "| before the tail: utilisation", round((ph["node_lanes"] - ph["tail_closest_lanes"] - ph["tail_shadow_lanes"]) / max(1, ph["node_slots"] - ph["tail_slots"]), 3))

if ph["waves"]:
    total = (ph["wave_end_max"] - ph["wave_start_min"]) / 100.0   # microseconds
    mean_end = (ph["wave_end_sum"] / ph["waves"] - ph["wave_start_min"]) / 100.0
    print(f"waves {ph['waves']}: kernel span {total:.0f} us, mean wave end at {mean_end:.0f} us -> the average wave idles {100 * (1 - mean_end / total):.1f} % of the span")
print("connect job loop: lane utilisation", round(ph["job_lanes"] / max(1, ph["job_slots"]), 3), "jobs per wave-round", ph["job_lanes"] / max(1, ph["job_slots"] // 64),
      "rounds", ph["job_slots"] // 64, "connections", cnt["connections"])
print("phase wave-clocks per wave-iteration-ish: shade", ph["shade"], "connect", ph["connect"], "pool", ph["shadow_pool"])
