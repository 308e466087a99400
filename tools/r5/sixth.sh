cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 300 bash tools/ab_pmc.sh ab_build/new.so ab_build/aux.so > gpurun_out/r5/abpmc5.txt 2>&1; grep "wave cycles" gpurun_out/r5/abpmc5.txt
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gputests6.log 2>&1; tail -12 gpurun_out/r5/gputests6.log
