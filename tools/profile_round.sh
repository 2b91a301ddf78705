#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs) of the bench command.
# Outputs land in gpurun_out/prof_<tag>/ ; tools/pmc_summary.py condenses them into profiles/.
TAG=${1:-r01}
# steps and warm-up are multiples of 32 so that every batched eye launch holds the 32 frames of bench.py's default
# BENCH_ARGS: extra bench.py arguments (e.g. --no-light-ahead for a run whose kernels never overlap)
BENCH_ARGS=${BENCH_ARGS:-}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 64 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 $BENCH_ARGS > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 $BENCH_ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 $BENCH_ARGS > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 $BENCH_ARGS > $OUT/bench_l2.json 2> $OUT/l2.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0 $BENCH_ARGS > $OUT/bench_sq.json 2> $OUT/sq.log
# round 4: direct unit-busy counters (which unit is nearest saturation?) -- each group its own pass, never with a trace domain beyond --kernel-trace
PASS_ARGS="--steps 32 --warmup 32 --no-cpu-baseline --long-steps 0 --sync-each-frames 0"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.log
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_sq3.json 2> $OUT/sq3.log
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_ta -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_ta.json 2> $OUT/ta.log
rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --kernel-trace --output-format csv -d $OUT/pmc_ta2 -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_ta2.json 2> $OUT/ta2.log
rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum --kernel-trace --output-format csv -d $OUT/pmc_ta3 -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_ta3.json 2> $OUT/ta3.log
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcp -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_tcp.json 2> $OUT/tcp.log
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/pmc_tcp2 -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_tcp2.json 2> $OUT/tcp2.log
rocprofv3 --pmc TD_TD_BUSY_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $OUT/pmc_td -- python3 $R/bench.py $PASS_ARGS $BENCH_ARGS > $OUT/bench_td.json 2> $OUT/td.log
find $OUT -name "*.csv" | head -40
# keep only the small per-kernel summaries + counter rows of our kernels (the raw traces are large)
for d in pmc_fetch pmc_write pmc_l2 pmc_sq pmc_sq2 pmc_sq3 pmc_ta pmc_ta2 pmc_ta3 pmc_tcp pmc_tcp2 pmc_td; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E "k_spcbpt|k_light_trace|k_lvc_compact" $f) > $OUT/${d}_ours.csv
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT
