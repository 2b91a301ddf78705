mkdir -p gpurun_out
TAG=${1:-r06d}
timeout -k 10 1000 bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_log.txt 2>&1
python tools/pmc_summary.py $TAG > gpurun_out/${TAG}_pmc_summary_log.txt 2>&1; tail -3 gpurun_out/${TAG}_pmc_summary_log.txt
cp profiles/${TAG}_* profiles/traffic_latest.json gpurun_out/ 2>/dev/null
timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --fast-math-line 0 --sync-each-frames 0 > gpurun_out/${TAG}_bench_default.json 2>gpurun_out/${TAG}_default_err.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --light-geometry reference --no-cpu-baseline --fast-math-line 0 --sync-each-frames 0 --long-steps 0 > gpurun_out/${TAG}_bench_refgeo.json 2> gpurun_out/${TAG}_refgeo_err.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench_err.log
python - $TAG <<'P'
import json,sys
tag=sys.argv[1]
for f in (f"{tag}_bench_default",f"{tag}_bench_refgeo",f"{tag}_bench"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d.get("ms_per_step_long"), "kernel", r["kernel_ms"], "frac", r["frac"], r.get("hbm_measured_frac"), d.get("ms_per_frame_sync_each"), d.get("ms_per_frame_viewer"), d.get("ms_per_frame_viewer_moving"), d.get("fast_math_build",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
P
