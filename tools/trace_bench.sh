#!/bin/bash
# developer: kernel timeline of bench.py with the given arguments (rocprofv3 --kernel-trace): the eye / light kernels and the
# sampler-build spans of the last ~150 ms
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tb
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tb -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/tb.json 2> $R/gpurun_out/tb.err
cd $R
python3 - <<'PY'
import csv, glob, json
rows = []
for f in glob.glob("gpurun_out/tb/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spc::", "")[:32], r.get("Stream_Id", "")))
rows.sort()
end = rows[-1][1]
sel = [r for r in rows if r[1] > end - 260e6]
t0 = sel[0][0]
prev_cmf = None
for s, e, n, st in sel:
    if "k_spcbpt" in n or "k_light_trace" in n or "k_film_merge" in n and False:
        print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}  ({(e - s) / 1e6:8.3f} ms)  {n}  s{st}")
    if n.startswith("k_cmf"):
        print(f"{(s - t0) / 1e6:9.3f}  sampler built (k_cmf)  s{st}" + (f"   +{(s - prev_cmf) / 1e6:.3f}" if prev_cmf else "")); prev_cmf = s
d = json.loads(open("gpurun_out/tb.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
rm -rf $R/gpurun_out/tb
