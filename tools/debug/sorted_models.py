"""Experiment: does it pay to put the slow randomised robots into waves of their own?  Per-env cost measured with one env per wave, the
SAME models then stepped unsorted and sorted by that cost (offsets -> randomise_models(offsets=...)), per-step launches, both kernels."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((400, n), generator=g, device=dev) * 2 - 1

def run(env, steps0, steps1, clocks=False):
    obs = torch.zeros((n, env.obs_dim), device=dev); rew = torch.zeros((n,), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
    env.reset_device()
    acc = None
    for k in range(steps0):
        env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps0, steps1):
        env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        if clocks and k % 5 == 0:
            torch.cuda.synchronize(dev); w = env.wave_clocks(); acc = w if acc is None else acc + w
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / (steps1 - steps0), acc

e1 = JitterbugVecEnv(n, "move_to_pose", seed=0, envs_per_wave=1, stream=torch.cuda.current_stream(dev).cuda_stream)
r = e1.randomise_models(seed=1000, return_params=False, return_offsets=True)
off = r["offsets"]
_, cost = run(e1, 50, 250, clocks=True)
e1.close()
print("per-env cost (one env per wave): mean %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f (relative to the median)" % tuple(x / np.median(cost) for x in (cost.mean(), np.median(cost), np.percentile(cost, 90), np.percentile(cost, 99), cost.max())))
perm = np.argsort(cost)
for name, o in (("unsorted", off), ("sorted by cost", off[perm]), ("unsorted", off), ("sorted by cost", off[perm])):
    for variant in ("auto", "ordinary"):
        env = JitterbugVecEnv(n, "move_to_pose", seed=0, variant=variant, per_env_model=True, stream=torch.cuda.current_stream(dev).cuda_stream)
        env.randomise_models(seed=1000, offsets=o, return_params=False)
        t, _ = run(env, 100, 400)
        print("%-16s %-10s %.3f ms/step  %.2f M env-steps/s" % (name, env.kernel_variant, t * 1e3, n / t / 1e6))
        env.close()
