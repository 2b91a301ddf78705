#!/bin/bash
# developer: per-kernel time of the wavefront eye pass (SPCBPT_EYE_PASS=wavefront) on the bench scene, by bounce range
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/wf
cd /tmp
SPCBPT_EYE_PASS=wavefront rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wf -- python3 $R/tools/wf_prof.py --trained > $R/gpurun_out/wf.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/wf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spc::", "")[:28]))
rows.sort()
# the last frame: from the last k_wf_gen on
last = max(i for i, r in enumerate(rows) if "k_wf_gen" in r[2])
fr = rows[last:]
t0 = fr[0][0]
agg = collections.OrderedDict()
seen = collections.Counter()
for s, e, n in fr:
    if "k_wf" not in n: continue
    seen[n] += 1
    b = seen[n]           # n-th launch of this kernel = bounce
    key = (n, "b1-3" if b <= 3 else ("b4-8" if b <= 8 else "b9+"))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
for k, v in agg.items(): print(f"{k[0]:30s} {k[1]:5s} launches {v[0]:3d}  {v[1]:8.3f} ms")
print("frame span", (fr[-1][1] - t0) / 1e6, "ms; kernel time", sum((e - s) for s, e, n in fr) / 1e6)
PY
rm -rf $R/gpurun_out/wf
