#!/bin/bash
# developer: one PMC pass over bench.py for k_spcbpt<false>; usage: pmc_quick.sh "COUNTER1 COUNTER2 ..."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${2:-} > /dev/null 2> $OUT/p.log
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_spcbpt<false, false>" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, "%.4g" % (sum(v) / len(v)), "n", len(v))
PY
rm -rf $OUT/p
