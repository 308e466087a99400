"""Fused K-step rollout against the step-by-step loop on the same workload (BASELINE configs[2] unless told otherwise).

    python tools/rollout_bench.py [--envs 4096] [--actions uniform|const1|policy] [--flags 0] [--augmented]

Prints env-steps/s of: 1000 single-step launches, fused launches of K = 20 (steps 5-25), 100 (x10) and 1000, and the per-wave clocks of
the K = 1000 launch (mean against slowest wave: what bounds a fused rollout)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--actions", default="uniform")
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--task", default="move_from_origin")
ap.add_argument("--augmented", action="store_true")
ap.add_argument("--envs-per-wave", type=int, default=0)
ap.add_argument("--brief", action="store_true", help="only the step-by-step loop and the K = 100 x 10 launches (A/B runs)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
n, K = args.envs, 1000


def fresh():
    env = JitterbugVecEnv(n, args.task, seed=0, flags=args.flags, envs_per_wave=args.envs_per_wave, stream=torch.cuda.current_stream(dev).cuda_stream)
    if args.augmented:
        env.randomise_models(seed=1000, return_params=False)
    env.reset_device()
    return env


g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
if args.actions == "const1":
    tape.fill_(1.0)
policy = args.actions == "policy"
D = fresh().obs_dim
obs = torch.zeros((n, D), device=dev); rew = torch.zeros((K, n), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8); act = torch.zeros((n,), device=dev)


def prime(env):
    """--brief: one step and a reset before the measured episode, on EVERY env of this run alike (the first launch of a handle has no
    wave clocks to order its waves by; the reset after a step starts the NEXT episode, so all the runs must do it to compare bits)"""
    if args.brief:
        env.step_many_device(1, None if policy else tape.data_ptr())
        env.reset_device()
    return env


def timed(fn):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


env = prime(fresh())
print("kernel variant %s, %d envs per wave, actions %s" % (env.kernel_variant, env.envs_per_wave, args.actions))


def loop():
    if policy:
        env.observe_device(obs.data_ptr())
    for k in range(K):
        if policy:
            env.policy_device(obs.data_ptr(), act.data_ptr())
            env.step_device(act.data_ptr(), obs.data_ptr(), rew[k].data_ptr(), done.data_ptr())
        else:
            env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew[k].data_ptr(), done.data_ptr())


t = timed(loop)
print("step by step, 1000 launches (full episode):   %10.0f env-steps/s   %.4f ms/step" % (n * K / t, 1e3 * t / K))
ref_rew = rew.clone()
env.close()

env = prime(fresh())
t = timed(lambda: env.step_many_device(K, None if policy else tape.data_ptr(), rewards_ptr=rew.data_ptr(), obs_last_ptr=obs.data_ptr(), done_last_ptr=done.data_ptr()))
wc = env.wave_clocks()
print("fused K = 1000 (full episode, one launch):    %10.0f env-steps/s   %.4f ms/step   bit-identical rewards: %s" % (n * K / t, 1e3 * t / K, bool(torch.equal(rew, ref_rew))))
print("    per-wave clocks of that launch: mean %.1f ms, median %.1f, p99 %.1f, slowest %.1f ms; mean/slowest %.3f; ceiling if bounded by the mean wave: %.0f env-steps/s" % (
    1e3 * wc.mean(), 1e3 * np.median(wc), 1e3 * np.percentile(wc, 99), 1e3 * wc.max(), wc.mean() / wc.max(), n * K / wc.mean()))
env.close()

env = prime(fresh())
def chunks(c):
    for k0 in range(0, K, c):
        env.step_many_device(c, None if policy else tape[k0:k0 + c].data_ptr(), rewards_ptr=rew[k0:k0 + c].data_ptr(), obs_last_ptr=obs.data_ptr(), done_last_ptr=done.data_ptr())
t = timed(lambda: chunks(100))
print("fused K = 100 x 10 launches (full episode):   %10.0f env-steps/s   %.4f ms/step   bit-identical rewards: %s" % (n * K / t, 1e3 * t / K, bool(torch.equal(rew, ref_rew))))
env.close()

if args.brief:
    sys.exit(0)
env = prime(fresh())
env.step_many_device(5, None if policy else tape[:5].data_ptr())
t = timed(lambda: env.step_many_device(20, None if policy else tape[5:25].data_ptr(), rewards_ptr=rew[5:25].data_ptr()))
print("fused K = 20 (steps 5-25, the driver's window): %8.0f env-steps/s   %.4f ms/step" % (n * 20 / t, 1e3 * t / 20))
env.step_many_device(75, None if policy else tape[25:100].data_ptr())
t = timed(lambda: env.step_many_device(300, None if policy else tape[100:400].data_ptr(), rewards_ptr=rew[100:400].data_ptr()))
wc = env.wave_clocks()
print("fused K = 300 (steps 100-400, steady window):  %9.0f env-steps/s   %.4f ms/step   mean/slowest wave %.3f" % (n * 300 / t, 1e3 * t / 300, wc.mean() / wc.max()))
env.close()
