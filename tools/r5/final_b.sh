# Round 5 end-of-round collection, part B: bench lines, soak, parity sweep, config-5 sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
echo "[final] bench (driver window)"; timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f/bench_driver_window.json 2> gpurun_out/r5f/bench_driver_window.err; tail -c 300 gpurun_out/r5f/bench_driver_window.json
echo "[final] bench (default)"; timeout -k 10 600 python bench.py > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; tail -c 300 gpurun_out/r5f/bench.json
echo "[final] soak"; timeout -k 10 500 python tools/soak_rollout.py 12 2>&1 | grep -v amdgpu > gpurun_out/r5f/soak.txt; cat gpurun_out/r5f/soak.txt
echo "[final] parity sweep"; timeout -k 10 600 python tools/parity_sweep.py 256 3 2>&1 | grep -v amdgpu > gpurun_out/r5f/parity_sweep.txt; tail -4 gpurun_out/r5f/parity_sweep.txt
echo "[final] sizes"; bash tools/r5/ab_sizes.sh jitterbug_amd/libjitterbug_hip.so 2>&1 | tee gpurun_out/r5f/sizes.txt
