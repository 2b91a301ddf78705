#!/bin/bash
# developer (GPU box): which PC-sampling configurations does this box accept?
export TMPDIR=/tmp; cd /tmp; export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 --list-avail 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -60
cat > /tmp/tiny.py <<'P'
import torch
x = torch.ones(1 << 24, device="cuda")
for _ in range(300): x.mul_(1.0001)
torch.cuda.synchronize()
P
for cfg in "host_trap time 1" "host_trap time 100" "host_trap time 10000" "stochastic cycles 65536" "stochastic cycles 1048576"; do
  set -- $cfg
  rm -rf /tmp/pcp; timeout -k 5 120 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d /tmp/pcp -- python3 /tmp/tiny.py > /tmp/pcp.log 2>&1
  echo "== $cfg: rc $? files: $(find /tmp/pcp -name '*pc_sampl*' | wc -l) $(grep -m1 -i 'not supported\|error' /tmp/pcp.log | cut -c1-160)"
  f=$(find /tmp/pcp -name '*pc_sampl*csv' | head -1); [ -n "$f" ] && (head -3 "$f"; wc -l "$f")
done
