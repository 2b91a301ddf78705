#!/bin/bash
# A/B on ONE box (boxes differ by a few percent): bench.py with the library in the tree (B) and with .ab/libA.so (A), alternating.
# usage (on the GPU box): bash tools/ab_bench.sh [bench args]
mkdir -p gpurun_out
for k in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export SPCBPT_LIB=$PWD/.ab/libA.so; else unset SPCBPT_LIB; fi
    python bench.py --no-cpu-baseline --fast-math-line 0 "$@" 2>gpurun_out/ab_err_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, kernel', d['roofline']['kernel_ms'])" || tail -5 gpurun_out/ab_err_$v.log
  done
done
