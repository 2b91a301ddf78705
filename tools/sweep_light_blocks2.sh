#!/bin/bash
# developer: grid of the batched light pass (SPCBPT_LIGHT_BATCH_BLOCKS) against the long steady-state run of bench.py
for lb in 24 32 40 48 64; do
  for k in 1 2; do
    SPCBPT_LIGHT_BATCH_BLOCKS=$lb python bench.py --no-cpu-baseline --steps 32 --warmup 4 --sync-each-frames 0 --long-steps 256 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('light blocks $lb:', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, long', d['ms_per_step_long'], 'kernel', d['roofline']['kernel_ms'], 'light', d['kernels_ms']['light_trace'])"
  done
done
