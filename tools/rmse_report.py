"""Equal-spp and equal-time RMSE (second half of the BASELINE metric): tools/rmse_lib.py on one of the BASELINE scenes.
  python tools/rmse_report.py <tag> [cornell|bedroom|hallway] [W H spp]   ->  profiles/<tag>_rmse_<scene>.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPCBPT_EYE_BATCH", "4")   # the timing loop batches eye launches like bench.py
import __graft_entry__ as g
from rmse_lib import rmse_study
p = g.load_package()
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "cornell"
W, H, N = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1024, 1024, 64)
scene = {"cornell": p.scenes.cornell_box, "bedroom": p.scenes.bedroom, "hallway": p.scenes.hallway}[name]()
out = rmse_study(p, scene, W, H, N, 400_000 if name == "cornell" else 2_000_000)
out["tag"] = tag
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_rmse_{name}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
