#!/bin/bash
# kernel timeline of the interactive loop (tools/viewer_loop.py): the last frames of mode 2, then of mode 0
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/timeline_viewer
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/viewer_loop.py > $OUT/out.txt 2> $OUT/log.txt
cat $OUT/out.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
eye=[i for i,r in enumerate(rows) if "k_spcbpt<false, false" in r["Kernel_Name"]]
def show(lo,hi,title):
    print(title)
    t0=int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi]:
        x=int(r["Start_Timestamp"]); z=int(r["End_Timestamp"])
        print(f"  {(x-t0)/1e3:9.1f} us  +{(z-x)/1e3:8.1f} us  {r['Kernel_Name'][:64]}")
# mode 2: eye kernels 4..15 ; mode 0: 20..31 (4 warm-up + 12 each)
show(eye[12]-2, eye[15]+6, "---- mode 2 (three frames)")
show(eye[28]-8, eye[31]+4, "---- mode 0 (three frames)")
P
find $OUT -name "*kernel_trace.csv" -delete
