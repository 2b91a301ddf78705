#!/bin/bash
# rocprofv3 PMC passes over tools/quad_trace_report.py (lane / quad / quad_x2 / quad_x4 on the three ray sets): which unit each form loads.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_quad
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/quad_trace_report.py --repeat 3 > $OUT/report.txt 2> $OUT/stats.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/quad_trace_report.py --repeat 1 > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/pmc_ta -- python3 $R/tools/quad_trace_report.py --repeat 1 > /dev/null 2> $OUT/ta.log
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcp -- python3 $R/tools/quad_trace_report.py --repeat 1 > /dev/null 2> $OUT/tcp.log
for d in pmc_sq pmc_ta pmc_tcp; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E "k_trace_" $f) > $OUT/${d}_ours.csv
done
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT
