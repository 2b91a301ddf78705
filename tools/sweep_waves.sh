#!/bin/bash
# developer sweep: rebuild the megakernel with different register budgets and bench each (run on the GPU box)
cd "$GRAFT_REPO_ROOT/spcbpt-optix7_amd/csrc"
for w in 2 3 4 5; do
  rm -f kernels.o libspcbpt_hip.so
  make EXTRA="-DSPC_WAVES=$w" > /dev/null 2>&1
  /opt/rocm/bin/hipcc -DSPC_WAVES=$w -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c kernels.hip -o /tmp/k.o 2>&1 | grep -A8 "k_spcbptILb0" | grep -E "VGPRs:|ScratchSize|Occupancy" | tr '\n' ' '
  echo
  (cd "$GRAFT_REPO_ROOT" && python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves=$w', d['value'], d['kernels_ms'])")
done
