# Where does the two-waves-per-SIMD variant start to pay (per-step launches)?  sizes between one and two waves per SIMD, both variants
run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-70s %10.0f env-steps/s  %.4f ms/step  %s finite %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['kernel_variant'], d['finite']))" "$@"; }
for n in 4608 5120 6144 7168 8192; do
run --envs-per-gpu $n --no-lean
run --envs-per-gpu $n --lean
done
for n in 4096 6144 8192 12288; do
run --augmented --task move_to_pose --envs-per-gpu $n --no-lean
run --augmented --task move_to_pose --envs-per-gpu $n --lean
done
