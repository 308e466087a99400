cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gputests9.log 2>&1; grep -v "^$" gpurun_out/r5/gputests9.log | tail -25
bash tools/r5/ab_sizes.sh ab_build/thread.so ab_build/thread2.so 2>&1 | grep augmented | tee gpurun_out/r5/absizes_thread2.txt
