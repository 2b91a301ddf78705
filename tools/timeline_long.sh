#!/bin/bash
# kernel timeline of a long bench run (several batched eye launches back to back): what sits between consecutive eye kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/timeline_long
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 128 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 > $OUT/bench.json 2> $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
big=[r for r in rows if "k_spcbpt<false, true" in r["Kernel_Name"]]
big=big[-5:]
for a,b in zip(big[:-1],big[1:]):
    e=int(a["End_Timestamp"]); s=int(b["Start_Timestamp"])
    print(f"eye kernel {(int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e6:.3f} ms; gap to the next {(s-e)/1e3:.1f} us:")
    for r in rows:
        x=int(r["Start_Timestamp"]); z=int(r["End_Timestamp"])
        if z>=e-200_000 and x<=s+50_000 and "k_spcbpt" not in r["Kernel_Name"]:
            print(f"    {(x-e)/1e3:9.1f} us  +{(z-x)/1e3:8.1f} us  {r['Kernel_Name'][:70]}")
P
find $OUT -name "*kernel_trace.csv" -delete
