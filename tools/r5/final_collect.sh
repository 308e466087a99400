# Round 5 end-of-round collection on one MI355X box: rocprofv3 kernel stats + PMC (per-step path and one fused launch), bench lines, soak, parity sweep.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
rm -rf gpurun_out/prof gpurun_out/prof_rollout
echo "[final] per-step profile"; timeout -k 10 600 bash tools/collect_profiles.sh 300 100 > gpurun_out/r5f/collect.log 2>&1; tail -2 gpurun_out/r5f/collect.log
echo "[final] fused profile"; timeout -k 10 400 bash tools/collect_rollout_profile.sh 1000 > gpurun_out/r5f/collect_rollout.log 2>&1; tail -2 gpurun_out/r5f/collect_rollout.log
find gpurun_out/prof gpurun_out/prof_rollout -name "*kernel_trace.csv" -size +3M -delete; find gpurun_out/prof gpurun_out/prof_rollout -name "*.db" -delete
echo "[final] bench (driver window)"; timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f/bench_driver_window.json 2> gpurun_out/r5f/bench_driver_window.err; tail -c 300 gpurun_out/r5f/bench_driver_window.json
echo "[final] bench (default)"; timeout -k 10 600 python bench.py > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; tail -c 300 gpurun_out/r5f/bench.json
echo "[final] soak"; timeout -k 10 500 python tools/soak_rollout.py 12 2>&1 | grep -v amdgpu > gpurun_out/r5f/soak.txt; cat gpurun_out/r5f/soak.txt
echo "[final] parity sweep"; timeout -k 10 600 python tools/parity_sweep.py 256 3 2>&1 | grep -v amdgpu > gpurun_out/r5f/parity_sweep.txt; tail -4 gpurun_out/r5f/parity_sweep.txt
