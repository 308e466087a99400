cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
JITTERBUG_HIP_LIB=ab_build/cap.so timeout -k 10 300 python tools/r5/harvest.py gpurun_out/r5/captured2.npy 8 > gpurun_out/r5/harvest2.txt 2>&1
cat gpurun_out/r5/harvest2.txt
timeout -k 10 400 python tools/soak_rollout.py 12 > gpurun_out/r5/soak2.txt 2>&1
cat gpurun_out/r5/soak2.txt
timeout -k 10 400 bash tools/ab_pmc.sh ab_build/base.so ab_build/new_mn8.so ab_build/new.so ab_build/new_mn16.so > gpurun_out/r5/abpmc2.txt 2>&1
cat gpurun_out/r5/abpmc2.txt
timeout -k 10 600 python tools/parity_sweep.py 256 1 > gpurun_out/r5/parity_sweep2.txt 2>&1
cat gpurun_out/r5/parity_sweep2.txt
