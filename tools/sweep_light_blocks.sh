for v in - 16 24 28 32; do
  if [ "$v" = "-" ]; then unset SPCBPT_LIGHT_BATCH_BLOCKS; else export SPCBPT_LIGHT_BATCH_BLOCKS=$v; fi
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --long-steps 256 --sync-each-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LIGHT_BATCH_BLOCKS=$v', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, long', d['ms_per_step_long'])"
done
