#!/bin/bash
for k in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export SPCBPT_LIB=$PWD/.ab/libA.so; else unset SPCBPT_LIB; fi
    python bench.py --no-cpu-baseline --steps 32 --warmup 4 --sync-each-frames 0 --long-steps 256 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, long', d['ms_per_step_long'], 'kernel', d['roofline']['kernel_ms'], 'build', d['kernels_ms']['sampler_build'])"
  done
done
