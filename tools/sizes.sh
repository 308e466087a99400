#!/bin/bash
# Throughput of the current build at the other BASELINE sizes / modes (run on the GPU box): tools/sizes.sh > gpurun_out/r03_sizes.txt
# (the product picks the LEAN kernel variant above 4096 envs per GPU, with one model per env from 8192 - jitterbug_amd.variants; --no-lean is the one-wave kernel on the same batch)
# Second part: the same sizes as fused rollouts (tools/rollout_bench.py --brief: step by step / one 1000-step launch / ten 100-step launches).
run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-70s %10.0f env-steps/s  %.4f ms/step  finite %s' % (' '.join(sys.argv[1:]) or '(default: move_from_origin N=4096)', d['value'], d['ms_per_step'], d['finite']))" "$@"; }
run
run --contacts 0
run --task move_to_pose
run --envs-per-gpu 6144 --no-lean
run --envs-per-gpu 6144
run --envs-per-gpu 8192 --no-lean
run --envs-per-gpu 8192
run --envs-per-gpu 16384 --no-lean
run --envs-per-gpu 16384
run --envs-per-gpu 65536 --no-lean
run --envs-per-gpu 65536
run --task move_to_pose --envs-per-gpu 32768 --no-lean
run --task move_to_pose --envs-per-gpu 32768
run --augmented --envs-per-gpu 8192 --task move_to_pose
run --augmented --envs-per-gpu 8192 --task move_to_pose --no-lean
run --augmented --envs-per-gpu 8192 --task move_to_pose --no-pair
run --augmented --envs-per-gpu 4096
run --actions const1
JB_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-70s %10.0f env-steps/s  %.4f ms/step  finite %s' % ('N>1 code path, one-rank RCCL group (ShardedJitterbugEnv depth 2)', d['value'], d['ms_per_step'], d['finite']))"

echo "# fused rollouts (jb_step_many_device), whole episode from the reset: step-by-step launches / one K=1000 launch / ten K=100 launches"
rb() { echo "## $*"; python tools/rollout_bench.py --brief "$@" 2>/dev/null | grep "kernel variant\|step by step\|fused K = 1000\|fused K = 100 "; }
rb --envs 4096
rb --envs 4096 --actions const1
rb --envs 4096 --actions policy
rb --envs 8192 --flags 2
rb --envs 16384 --flags 2
rb --envs 65536 --flags 2
rb --envs 32768 --flags 2 --task move_to_pose
rb --envs 8192 --augmented --task move_to_pose
rb --envs 8192 --augmented --task move_to_pose --flags 2
rb --envs 16384 --augmented --task move_to_pose --flags 2
