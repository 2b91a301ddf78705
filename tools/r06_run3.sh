mkdir -p gpurun_out
timeout -k 10 500 bash tools/ab_variants.sh --steps 64 --warmup 32 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06_probes2.txt 2>&1
cat gpurun_out/r06_probes2.txt
timeout -k 10 500 bash tools/sweep_env.sh SPCBPT_BVH_NODE_COST "0.5 - 2" --steps 64 --warmup 32 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06_sweep_node_cost.txt 2>&1
cat gpurun_out/r06_sweep_node_cost.txt
