mkdir -p gpurun_out
for v in lanepairs TREE lanepairs TREE; do
  if [ $v = TREE ]; then unset SPCBPT_LIB; else export SPCBPT_LIB=$PWD/.ab/lib$v.so; fi
  echo "$v: $(timeout -k 10 200 python tools/aux_kernels_time.py 2>>gpurun_out/r06_aux_err.log | tr '\n' ' ')"
done > gpurun_out/r06_ab_lane_pairs.txt
cat gpurun_out/r06_ab_lane_pairs.txt
