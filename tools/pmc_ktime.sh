#!/bin/bash
# developer: one PMC pass over tools/ktime.py, counters of the plain and the counting megakernel side by side
# usage: pmc_ktime.sh "COUNTER1 COUNTER2 ..." [ktime args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ktime
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/p -- python3 $R/tools/ktime.py ${2:-} > $OUT/ktime.log 2> $OUT/p.log
cat $OUT/ktime.log
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    for tag in ("k_spcbpt<false, false>", "k_spcbpt<true, false>"):
        if tag in r["Kernel_Name"]:
            agg[(r["Counter_Name"], tag)].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], "%.5g" % (sum(v) / len(v)), "n", len(v))
PY
rm -rf $OUT/p
