#!/bin/bash
# kernel timeline of the contract's bench run (rocprofv3 --kernel-trace): what runs between the launches of the timed region
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 > $OUT/bench.json 2> $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the timed region = around the LAST batched eye launch with 20 frames: find the longest k_spcbpt<false, true
big=[r for r in rows if "k_spcbpt<false, true" in r["Kernel_Name"]]
big.sort(key=lambda r:int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
b=big[-1]; s=int(b["Start_Timestamp"]); e=int(b["End_Timestamp"])
print("eye batch", (e-s)/1e6, "ms")
t0=s-6_000_000; t1=e+3_000_000
last=None
for r in rows:
    a=int(r["Start_Timestamp"]); z=int(r["End_Timestamp"])
    if a<t0 or a>t1: continue
    print(f"{(a-s)/1e3:10.1f} us  +{(z-a)/1e3:8.1f} us  {r['Kernel_Name'][:60]}")
P
find $OUT -name "*kernel_trace.csv" -delete
