set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5/tests1.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5/tests1.log
tail -5 gpurun_out/r5/tests1.log
timeout -k 10 300 bash tools/ab_pmc.sh ab_build/base.so ab_build/new_nols.so ab_build/new.so > gpurun_out/r5/abpmc1.txt 2>&1
cat gpurun_out/r5/abpmc1.txt
timeout -k 10 400 python tools/soak_rollout.py 6 > gpurun_out/r5/soak1.txt 2>&1
cat gpurun_out/r5/soak1.txt
