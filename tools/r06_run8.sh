mkdir -p gpurun_out
timeout -k 10 1100 bash tools/profile_round.sh r06c > gpurun_out/r06c_profile_log.txt 2>&1
tail -5 gpurun_out/r06c_profile_log.txt
python tools/pmc_summary.py r06c > gpurun_out/r06c_pmc_summary_log.txt 2>&1; tail -5 gpurun_out/r06c_pmc_summary_log.txt
cp profiles/r06c_* profiles/traffic_latest.json gpurun_out/ 2>/dev/null
