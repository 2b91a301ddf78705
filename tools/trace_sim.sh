#!/bin/bash
# developer: kernel timeline of tools/rank_sim.py with the given arguments (rocprofv3 --kernel-trace), last N rows
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/tools/rank_sim.py "$@" > $R/gpurun_out/tl.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spc::", "")[:40], r.get("Stream_Id", "")))
rows.sort()
t0 = rows[0][0]
for s, e, n, st in rows[-220:]:
    print(f"{(s - t0) / 1e6:10.3f} -> {(e - t0) / 1e6:10.3f}  ({(e - s) / 1e6:7.3f} ms)  {n}  s{st}")
coll = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower() or "k_gather_compact" in r[2] or "k_pack_shards" in r[2]]
eyes = [r for r in rows if "k_spcbpt" in r[2]]
import collections
print("---- exchange kernels in the trace:", dict(collections.Counter(r[2] for r in coll)), "| eye launches:", len(eyes))
print("---- eye / light kernels only")
for s, e, n, st in [r for r in rows if "k_spcbpt" in r[2] or "k_light_trace" in r[2]][-40:]:
    print(f"{(s - t0) / 1e6:10.3f} -> {(e - t0) / 1e6:10.3f}  ({(e - s) / 1e6:7.3f} ms)  {n}  s{st}")
PY
grep "N=" $R/gpurun_out/tl.log
rm -rf $R/gpurun_out/tl
