run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-60s %10.0f env-steps/s  %.4f ms/step  finite %s cap %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['finite'], d['solver_cap_hits']))" "$@"; }
for n in 4096 8192 16384 65536; do run --envs-per-gpu $n; run --envs-per-gpu $n --lean; run --envs-per-gpu $n --lean --envs-per-wave 4; done
run --envs-per-gpu 4096 --lean --envs-per-wave 1
run --augmented --envs-per-gpu 8192 --task move_to_pose --lean
run --actions const1 --lean
