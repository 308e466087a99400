cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_thread_contact.py tests/test_pair_contact.py tests/test_randomise.py "tests/test_gpu_clearance.py::test_no_geom_pair_ever_touches[augmented]" -q -m gpu -s > gpurun_out/r5/gputests8.log 2>&1; grep -v "^$" gpurun_out/r5/gputests8.log | tail -25
bash tools/r5/ab_sizes.sh ab_build/aux.so ab_build/thread.so 2>&1 | grep augmented | tee gpurun_out/r5/absizes_thread.txt
