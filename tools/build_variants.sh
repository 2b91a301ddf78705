#!/bin/bash
# developer: builds libspcbpt_hip.so variants into .ab/lib<name>.so (for tools/ab_variants.sh on the GPU box), each in a scratch copy
# of the sources so that the tree's own library is left alone.  usage: tools/build_variants.sh name1="-DSPC_X=1 -DSPC_Y=2" name2="..."
# (hipcc cross-compiles here; the libraries travel to the GPU box with the tree -- .ab/ is git-ignored, not gpurun-ignored)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/.ab"
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  (
    W=$(mktemp -d /tmp/spc_variant_XXXX)
    mkdir -p "$W/pkg" "$W/include"
    cp -r "$ROOT/spcbpt-optix7_amd/csrc" "$W/pkg/csrc"
    cp "$ROOT"/include/*.h "$W/include/"
    cd "$W/pkg/csrc"
    rm -f *.o *.so .hip_flags
    if make -j3 EXTRA="$flags" libspcbpt_hip.so > "$W/build.log" 2>&1; then
      cp libspcbpt_hip.so "$ROOT/.ab/lib$name.so"
      echo "built .ab/lib$name.so  [$flags]"
    else
      echo "FAILED $name: see $W/build.log"; tail -20 "$W/build.log"; exit 1
    fi
    rm -rf "$W"
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 3 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
