# A/B of two builds on one box: tools/debug/ab_sizes.sh libA.so libB.so ...   (JITTERBUG_HIP_LIB selects the library; sizes where the LEAN kernels run)
run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-70s %10.0f env-steps/s  %.4f ms/step  %s finite %s' % (' '.join(sys.argv[1:]) or '(default: move_from_origin N=4096)', d['value'], d['ms_per_step'], d['kernel_variant'], d['finite']))" "$@"; }
for lib in "$@"; do
echo "## $lib"
export JITTERBUG_HIP_LIB=$lib
run
run --envs-per-gpu 8192
run --envs-per-gpu 6144 --lean
run --envs-per-gpu 65536
run --augmented --envs-per-gpu 8192 --task move_to_pose
run --augmented --envs-per-gpu 8192 --task move_to_pose --lean
run --actions const1 --envs-per-gpu 8192
done
