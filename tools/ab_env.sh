#!/bin/bash
# A/B on ONE box of the library in the tree with (A) and without (B) one environment switch set at spcbpt_create, alternating.
# usage (on the GPU box): bash tools/ab_env.sh SPCBPT_NO_FAN_TAIL [bench args]
mkdir -p gpurun_out
sw=$1; shift
for k in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export $sw=1; else unset $sw; fi
    python bench.py --no-cpu-baseline --fast-math-line 0 "$@" 2>gpurun_out/ab_err_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, kernel', d['roofline']['kernel_ms'], 'viewer', d.get('ms_per_frame_viewer'))" || tail -5 gpurun_out/ab_err_$v.log
  done
done
