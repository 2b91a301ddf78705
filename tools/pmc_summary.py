"""Condenses the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / average ns)
  profiles/<tag>_pmc_summary.json   per-kernel averages of the PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss, SQ_*)
  profiles/traffic_latest.json      HBM-side bytes per k_spcbpt launch, read by bench.py for roofline.traffic
Usage: python tools/pmc_summary.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))   # the newest run
summary = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*_ours.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        summary.setdefault(k, {})[c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
for k, d in summary.items():
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        h, m = d["TCC_HIT_sum"]["avg_per_launch"], d["TCC_MISS_sum"]["avg_per_launch"]
        d["l2_hit_rate"] = h / (h + m)
    if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
        d["wait_any_frac"] = d["SQ_WAIT_ANY"]["avg_per_launch"] / d["SQ_WAVE_CYCLES"]["avg_per_launch"]
notes = ("FETCH_SIZE / WRITE_SIZE are in KB per launch, collected in separate --pmc passes (TCC slot limits). "
         "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads exactly half of a wide coalesced 16-B/lane stream and is "
         "uncalibrated for other patterns; this kernel issues scattered 16-B/lane gathers, so hbm bytes are reported as a "
         "range [FETCH_SIZE, 2*FETCH_SIZE] + WRITE_SIZE; Infinity-Cache hits are included in these memory-side counters.")
out = {"tag": tag, "notes": notes, "kernels": summary}
# the dominant kernel of the run: the batched instantiation when the run used it, else the single-frame one (non-counting)
key = next((k for k in ("spc::k_spcbpt<false, true, true, false>", "spc::k_spcbpt<false, true, true, true>", "spc::k_spcbpt<false, false, true, false>", "spc::k_spcbpt<false, true, true>", "spc::k_spcbpt<false, false, true>", "spc::k_spcbpt<false, true>", "spc::k_spcbpt<false, false>", "spc::k_spcbpt<false>")
            if k in summary and "FETCH_SIZE" in summary[k]), "")
frames_per_launch = int(os.environ.get("FRAMES_PER_LAUNCH", "32" if key.startswith("spc::k_spcbpt<false, true") else "1"))
if key in summary and "FETCH_SIZE" in summary[key] and "WRITE_SIZE" in summary[key]:
    f = summary[key]["FETCH_SIZE"]["avg_per_launch"] * 1024.0
    w = summary[key]["WRITE_SIZE"]["avg_per_launch"] * 1024.0
    out["spcbpt_render_hbm_bytes_per_launch_low"] = f + w
    out["spcbpt_render_hbm_bytes_per_launch_high"] = 2 * f + w
    import __graft_entry__ as g
    d = summary[key]
    valu = None
    if "SQ_INSTS_VALU" in d and "SQ_BUSY_CYCLES" in d:
        # wave-instructions x 4 cycles (one wave's issue cost, MI355X_MICROARCH.md) over the cycles of all 1024 SIMDs
        # (SQ_BUSY_CYCLES is summed over the 32 shader engines: x 32 SIMDs per engine)
        valu = d["SQ_INSTS_VALU"]["avg_per_launch"] * 4.0 / (d["SQ_BUSY_CYCLES"]["avg_per_launch"] * 32.0)
    json.dump({"tag": tag, "kernel": key, "frames_per_launch": frames_per_launch, "spcbpt_render_hbm_bytes_per_launch": 2 * f + w,
               "spcbpt_render_hbm_bytes_per_launch_low": f + w, "source_hash": g.load_package().api.source_hash(), "kernel_hash": g.load_package().api.kernel_hash(), "valu_issue_frac": valu,
               "definition": "2*FETCH_SIZE + WRITE_SIZE (KB*1024) per k_spcbpt<false> launch, gfx950 FETCH_SIZE half-count correction applied; "
                             "uncorrected lower bound = FETCH_SIZE + WRITE_SIZE = %.4g" % (f + w)},
              open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print("wrote", os.path.join(dst, f"{tag}_pmc_summary.json"))
