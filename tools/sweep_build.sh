#!/bin/bash
# developer sweep (GPU box): rebuild kernels.o with each EXTRA flag set given as arguments and bench
cd "$GRAFT_REPO_ROOT/spcbpt-optix7_amd/csrc"
for flags in "$@"; do
  rm -f kernels.o capi.o preprocess.o libspcbpt_hip.so
  make EXTRA="$flags" > /dev/null 2>&1
  /opt/rocm/bin/hipcc $flags -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c kernels.hip -o /tmp/k.o 2>&1 | grep -A8 "k_spcbptILb0" | grep -E "VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' '
  (cd "$GRAFT_REPO_ROOT" && python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', d['value'], d['kernels_ms'])")
done
