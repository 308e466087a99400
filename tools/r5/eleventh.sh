cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
run() { timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-62s %10.0f env-steps/s  %.4f ms/step  variant %s finite %s unconverged %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['kernel_variant'], d['finite'], d['solver_cap_hits']))" "$@"; }
for rep in 1 2; do
run --envs-per-gpu 4096
run --envs-per-gpu 4096 --lean --envs-per-wave 4
run --envs-per-gpu 4096 --lean --envs-per-wave 2
run --envs-per-gpu 4096 --lean --envs-per-wave 1
run --envs-per-gpu 4096 --envs-per-wave 2
done 2>&1 | tee gpurun_out/r5/epw_lean_4096.txt
