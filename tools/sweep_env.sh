#!/bin/bash
# bench.py under a list of settings of ONE environment switch, on one box.  usage: bash tools/sweep_env.sh VAR "v1 v2 v3" [bench args]; "-" = unset
var=$1; vals=$2; shift 2
for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python bench.py --no-cpu-baseline --fast-math-line 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step, kernel', d['roofline']['kernel_ms'])"
done
