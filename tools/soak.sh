#!/bin/bash
# developer: long runs that wrap the ring of buffer sets many times -- bench.py with 400 steps, and the C++ N-GPU driver (3 local
# ranks, 96 frames) with and without batched light passes: the two images must be identical byte for byte.
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import __graft_entry__ as g
p = g.load_package()
print(p.scenes.write_gltf(p.scenes.bedroom(target_tris=60000), 'gpurun_out/soak_scene', 'bedroom'))
PY
S=$(ls gpurun_out/soak_scene/*.gltf | head -1)
for lb in 0 1; do
  ./tools/spcbpt_render_mgpu $S --local 3 --dim 320x200 --frames 96 --light-batch $lb --no-train --out gpurun_out/soak_lb$lb.ppm || exit 1
done
cmp gpurun_out/soak_lb0.ppm gpurun_out/soak_lb1.ppm && echo "driver images identical"
./tools/spcbpt_render_mgpu $S --gpus 1 --dim 320x200 --frames 96 --no-train --out gpurun_out/soak_single.ppm && cmp gpurun_out/soak_lb0.ppm gpurun_out/soak_single.ppm && echo "3 local ranks == 1 GPU"
python3 bench.py --steps 400 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('400 steps:', d['value'], d['ms_per_step'])"
rm -rf gpurun_out/soak_scene gpurun_out/soak_*.ppm
