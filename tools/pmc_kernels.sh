#!/bin/bash
# developer: PMC passes over bench.py, averaged per kernel (every kernel whose name matches $KMATCH, default: the eye kernels).
# usage (GPU box): [SPCBPT_EYE_SPLIT=1] bash tools/pmc_kernels.sh tag [bench args]     -> gpurun_out/pmc_<tag>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/pmc_k_$TAG
mkdir -p $OUT
cd /tmp
i=0
: > $R/gpurun_out/pmc_$TAG.txt
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-cpu-baseline --long-steps 0 --sync-each-frames 0 "$@" > /dev/null 2> $OUT/p$i.log
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && KMATCH="${KMATCH:-k_spcbpt<false|k_eye_paths|k_shadow_rays|k_connect_jobs|k_film_resolve}" python3 - "$f" >> $R/gpurun_out/pmc_$TAG.txt <<'PY'
import csv, sys, collections, os, re
pat = re.compile(os.environ["KMATCH"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if pat.search(n):
        short = n.split("(")[0].replace("void ", "").replace("spc::", "")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    for name, v in c.items():
        print(f"{k:32s} {name:28s} {sum(v) / len(v):.5g}  (launches {len(v)})")
PY
  rm -rf $OUT/p$i
done
cat $R/gpurun_out/pmc_$TAG.txt
