cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
JITTERBUG_HIP_LIB=ab_build/cap.so timeout -k 10 500 python tools/r5/harvest.py gpurun_out/r5/captured.npy 8 > gpurun_out/r5/harvest.txt 2>&1
cat gpurun_out/r5/harvest.txt
