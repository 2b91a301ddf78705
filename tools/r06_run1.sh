mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests1.log
tail -3 gpurun_out/r06_gputests1.log
timeout -k 10 600 bash tools/ab_variants.sh --steps 64 --warmup 32 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06_probes1.txt 2>&1
cat gpurun_out/r06_probes1.txt
