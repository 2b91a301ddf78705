mkdir -p gpurun_out
timeout -k 10 700 bash tools/ab_variants.sh --steps 64 --warmup 32 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06_ab_tri_batch.txt 2>&1
cat gpurun_out/r06_ab_tri_batch.txt
for v in lanepairs TREE lanepairs TREE; do
  if [ $v = TREE ]; then unset SPCBPT_LIB; else export SPCBPT_LIB=$PWD/.ab/lib$v.so; fi
  echo "$v: $(timeout -k 10 200 python tools/aux_kernels_time.py 2>/dev/null | tr '\n' ' ')"
done > gpurun_out/r06_ab_lane_pairs.txt
cat gpurun_out/r06_ab_lane_pairs.txt
