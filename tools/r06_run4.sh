mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fast_build.py > gpurun_out/r06_gputests2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests2.log
tail -15 gpurun_out/r06_gputests2.log
timeout -k 10 500 bash tools/ab_variants.sh --steps 64 --warmup 32 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06_ab_pairs.txt 2>&1
cat gpurun_out/r06_ab_pairs.txt
