mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gputests3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests3.log
tail -6 gpurun_out/r06_gputests3.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench_err.log; python -c "
import json; d=json.loads(open('gpurun_out/r06b_bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','ms_per_step_long','ms_per_frame_sync_each','ms_per_frame_viewer','ms_per_frame_viewer_moving')}); print(d['roofline']['kernel_ms'], d.get('fast_math_build'))"
