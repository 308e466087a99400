"""Diagnostic: per-step launches of the LEAN + PAIR kernel at 8192 per-env models - launch time against the waves' lifetimes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for flags, name in ((2, "lean_pair, longest first"), (2 | 32, "lean_pair, index order"), (0, "pair"), (32, "pair, index order")):
    env = JitterbugVecEnv(n, "move_to_pose", seed=0, flags=flags, stream=torch.cuda.current_stream(dev).cuda_stream)
    env.randomise_models(seed=1000, return_params=False)
    env.reset_device()
    g = torch.Generator(device=dev); g.manual_seed(1234)
    tape = torch.rand((400, n), generator=g, device=dev) * 2 - 1
    obs = torch.zeros((n, env.obs_dim), device=dev); rew = torch.zeros((n,), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
    for k in range(100):
        env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    wcs = []
    for k in range(100, 400):
        env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        if k % 30 == 0:
            torch.cuda.synchronize(dev); wcs.append(env.wave_clocks() * 1e3)
    torch.cuda.synchronize(dev)
    t = (time.perf_counter() - t0) / 300
    w = np.stack(wcs)
    srt = np.sort(w, axis=1)
    print("%-28s %s: %.3f ms/step = %.2f M env-steps/s | wave life ms: mean %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f | sum of wave life / (1024 SIMDs x step) = %.2f" % (
        name, env.kernel_variant, t * 1e3, n / t / 1e6, w.mean(), np.median(w), srt[:, int(0.9 * w.shape[1])].mean(), srt[:, int(0.99 * w.shape[1])].mean(), srt[:, -1].mean(), w.sum(1).mean() / 1024 / (t * 1e3)))
    env.close()
