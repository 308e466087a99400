run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-70s %10.0f env-steps/s  %.4f ms/step  finite %s' % (' '.join(sys.argv[1:]) or '(default: move_from_origin N=4096)', d['value'], d['ms_per_step'], d['finite']))" "$@"; }
for lib in "$@"; do
echo "## $lib"
export JITTERBUG_HIP_LIB=$lib
run
run --envs-per-gpu 8192
run --envs-per-gpu 65536
run --augmented --envs-per-gpu 8192 --task move_to_pose
run --actions const1
done
