"""Diagnostic: wave lifetimes of single step launches in the driver's window (steps 5-25 of an episode), N = 4096."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
dev = torch.device("cuda", 0)
n = 4096
env = JitterbugVecEnv(n, "move_from_origin", seed=0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((400, n), generator=g, device=dev) * 2 - 1
obs = torch.zeros((n, env.obs_dim), device=dev); rew = torch.zeros((n,), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
env.reset_device()
prev = None
for k in range(400):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    e1.record()
    torch.cuda.synchronize(dev)
    if k in (5, 10, 15, 20, 24, 100, 200, 300):
        w = env.wave_clocks() * 1e3
        o = np.argsort(-w)
        same = "" if prev is None else " | of the 16 slowest waves, %d were among the 16 slowest at the previous print" % len(set(o[:16]) & set(prev))
        prev = o[:16]
        print("step %3d: launch %.3f ms | wave life mean %.3f median %.3f p99 %.3f max %.3f | mean/launch %.2f%s" % (k, e0.elapsed_time(e1), w.mean(), np.median(w), np.percentile(w, 99), w.max(), w.mean() / e0.elapsed_time(e1), same))
env.close()
