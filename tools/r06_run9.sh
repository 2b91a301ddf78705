mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fan_pairs.py tests/test_gpu_film_golden.py -q -x -m gpu > gpurun_out/r06_gputests4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests4.log; tail -5 gpurun_out/r06_gputests4.log
for n in 1 2 4 8; do
  timeout -k 10 300 python tools/rank_sim.py $n 1 64 --native --batch=32 --lbatch --xbatch --bbatch --trained 2>>gpurun_out/r06_rank_sim_err.log | tail -1
done > gpurun_out/r06_rank_sim.txt
cat gpurun_out/r06_rank_sim.txt
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --fast-math-line 0 --sync-each-frames 0 > gpurun_out/r06c_bench_default.json 2>gpurun_out/r06c_default_err.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --light-geometry reference --no-cpu-baseline --fast-math-line 0 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06c_bench_refgeo.json 2> gpurun_out/r06c_refgeo_err.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench_err.log
python - <<'P'
import json
for f in ("r06c_bench_default","r06c_bench_refgeo","r06c_bench"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d.get("ms_per_step_long"), "kernel", r["kernel_ms"], "frac", r["frac"], "traffic", r["traffic"], r.get("hbm_measured_frac"), d.get("fast_math_build",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
P
