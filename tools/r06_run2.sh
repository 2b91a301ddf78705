mkdir -p gpurun_out
for s in 1 2; do
  timeout -k 10 300 python tools/rank_sim.py 8 $s 128 --native --batch=32 --lbatch --xbatch --bbatch --trained 2>gpurun_out/r06_rank_sim_err_$s.log | tail -1
done > gpurun_out/r06_rank_sim_streams.txt
cat gpurun_out/r06_rank_sim_streams.txt
for w in 256 512 1024; do
  SPCBPT_LIGHT_BLOCKS_WIDE=$w timeout -k 10 300 python bench.py --steps 8 --warmup 2 --long-steps 0 --no-cpu-baseline --fast-math-line 0 2>gpurun_out/r06_wide_err_$w.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide $w', 'sync_each', d['ms_per_frame_sync_each'], 'light_ahead', d['ms_per_frame_sync_each_light_ahead'], 'viewer', d['ms_per_frame_viewer'], 'moving', d['ms_per_frame_viewer_moving'])"
done > gpurun_out/r06_wide_light.txt
cat gpurun_out/r06_wide_light.txt
timeout -k 10 300 python tools/tail_probe.py > gpurun_out/r06_tail_probe.txt 2>&1; tail -8 gpurun_out/r06_tail_probe.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench_err.log; python -c "
import json; d=json.loads(open('gpurun_out/r06a_bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','ms_per_step_long','ms_per_frame_sync_each','ms_per_frame_viewer','ms_per_frame_viewer_moving')}); print(d.get('cpu_baseline')); print(d.get('fast_math_build'))"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --light-geometry reference --no-cpu-baseline --fast-math-line 0 --sync-each-frames 0 --long-steps 0 > gpurun_out/r06_bench_refgeo.json 2> gpurun_out/r06_refgeo_err.log; python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_refgeo.json').read().strip().splitlines()[-1]); print('refgeo', d['value'], d['ms_per_step'], d['config']['workload'][-80:])"
