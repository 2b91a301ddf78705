#!/bin/bash
# developer (GPU box): PC-sampling profile of the bench run's eye megakernel (rocprofv3 beta feature; its own run, no counters)
# needs .ab/libglines.so = the library built with EXTRA=-gline-tables-only (tools/build_variants.sh glines="-gline-tables-only")
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/pcs_out; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
SPCBPT_LIB=$R/.ab/libglines.so timeout -k 10 500 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${PCS_INTERVAL:-1000} --output-format csv -d $OUT -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline --fast-math-line 0 --long-steps 0 --sync-each-frames 0 > $R/gpurun_out/r06_pcs_bench.json 2> $R/gpurun_out/r06_pcs_log.txt
echo "rocprofv3 rc $?"; tail -3 $R/gpurun_out/r06_pcs_log.txt
find $OUT -type f | head; du -sh $OUT
python3 $R/tools/pc_hist.py $OUT $R/.ab/libglines.so > $R/gpurun_out/r06_pc_hist.txt 2>&1
head -60 $R/gpurun_out/r06_pc_hist.txt
