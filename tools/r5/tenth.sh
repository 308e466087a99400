cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_thread_contact.py tests/test_pair_contact.py tests/test_randomise.py -q -m gpu -s > gpurun_out/r5/gputests11.log 2>&1; grep -v "^$" gpurun_out/r5/gputests11.log | tail -12
bash tools/r5/ab_sizes.sh ab_build/thread2.so ab_build/thread3.so ab_build/thread2.so ab_build/thread3.so 2>&1 | grep augmented | tee gpurun_out/r5/absizes_thread3.txt
