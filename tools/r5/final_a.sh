# Round 5 end-of-round collection, part A: the GPU suite, then rocprofv3 kernel stats + PMC (per-step path and one fused launch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
rm -rf gpurun_out/prof gpurun_out/prof_rollout
echo "[final] gpu suite"; timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r5f/gputests.log 2>&1; tail -3 gpurun_out/r5f/gputests.log
echo "[final] per-step profile"; timeout -k 10 600 bash tools/collect_profiles.sh 300 100 > gpurun_out/r5f/collect.log 2>&1; tail -2 gpurun_out/r5f/collect.log
echo "[final] fused profile"; timeout -k 10 400 bash tools/collect_rollout_profile.sh 1000 > gpurun_out/r5f/collect_rollout.log 2>&1; tail -2 gpurun_out/r5f/collect_rollout.log
find gpurun_out/prof gpurun_out/prof_rollout -name "*kernel_trace.csv" -size +3M -delete; find gpurun_out/prof gpurun_out/prof_rollout -name "*.db" -delete
