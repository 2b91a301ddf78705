#!/bin/bash
# rocprofv3 kernel stats of the DRIVER'S bench command (20 steps, 5 warm-up: one 20-frame eye launch in the timed region) -- the
# summary whose average k_spcbpt duration bench.py's roofline.kernel_ms (HIP events) must agree with.  usage (GPU box): bash tools/stats_driver_cmd.sh <tag>
TAG=${1:-r06d}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stats_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --fast-math-line 0 > $OUT/bench.json 2> $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_driver_cmd_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
head -4 $R/gpurun_out/${TAG}_driver_cmd_kernel_stats.csv | cut -c1-160
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench under rocprof:', d['value'], d['ms_per_step'], 'kernel_ms (HIP events, 20-frame launch)', d['roofline']['kernel_ms'], 'launches', d['roofline']['launches'])"
