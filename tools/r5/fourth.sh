cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 700 python -m pytest tests/test_gpu_surface.py tests/test_bench_launch.py -q -m gpu > gpurun_out/r5/gputests4.log 2>&1; tail -25 gpurun_out/r5/gputests4.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke.txt 2>&1; tail -3 gpurun_out/r5/smoke.txt
rm -rf gpurun_out/r5/hiptrace; mkdir -p gpurun_out/r5/hiptrace
( cd /tmp && export TMPDIR=/tmp && JB_MARK_FILE=$GRAFT_REPO_ROOT/gpurun_out/r5/marks.txt timeout -k 10 300 rocprofv3 --hip-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/hiptrace -- python3 $GRAFT_REPO_ROOT/tools/r5/rollout10.py > $GRAFT_REPO_ROOT/gpurun_out/r5/rollout10.txt 2>&1 )
tail -12 gpurun_out/r5/rollout10.txt
python3 tools/r5/rollout_alloc_count.py gpurun_out/r5/hiptrace gpurun_out/r5/marks.txt 2>&1 | tee gpurun_out/r5/alloc_count.txt
find gpurun_out/r5/hiptrace -name "*.csv" -size +20M -delete
