mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gputests5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests5.log
tail -6 gpurun_out/r06_gputests5.log
