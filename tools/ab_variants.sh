#!/bin/bash
# developer: bench.py (+ the counting build's tail statistics) with every library under .ab/ and with the one in the tree, on ONE box
# usage (on the GPU box): bash tools/ab_variants.sh [--stats] [bench args]
STATS=0; if [ "$1" = "--stats" ]; then STATS=1; shift; fi
for lib in .ab/lib*.so TREE; do
  if [ $lib = TREE ]; then unset SPCBPT_LIB; else export SPCBPT_LIB=$PWD/$lib; fi
  for k in 1 2; do
    python bench.py --no-cpu-baseline --fast-math-line 0 "$@" 2>>gpurun_out/ab_variants_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, kernel', d['roofline']['kernel_ms'])"
  done
  if [ $STATS = 1 ]; then python tools/phase_check.py --trained 2>/dev/null | grep -E "node-loop|pooled|donation"; fi
done
