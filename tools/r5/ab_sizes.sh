# tools/r5/ab_sizes.sh lib1.so lib2.so ...: the LEAN / PAIR sizes for each build on one box (per-step launches, steps 50-350)
cd $GRAFT_REPO_ROOT
run() { lib=$1; shift; JITTERBUG_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-22s %-62s %10.0f env-steps/s  %.4f ms/step  variant %s finite %s unconverged %s' % (sys.argv[1], ' '.join(sys.argv[2:]), d['value'], d['ms_per_step'], d['kernel_variant'], d['finite'], d['solver_cap_hits']))" $lib "$@"; }
for lib in "$@"; do
  run $lib --envs-per-gpu 8192
  run $lib --envs-per-gpu 65536
  run $lib --task move_to_pose --envs-per-gpu 32768
  run $lib --augmented --envs-per-gpu 8192 --task move_to_pose
  run $lib --augmented --envs-per-gpu 8192 --task move_to_pose --no-lean
  run $lib --augmented --envs-per-gpu 16384 --task move_to_pose
done
