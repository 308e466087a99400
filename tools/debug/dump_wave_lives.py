"""Diagnostic: wave lifetimes of consecutive per-step launches (one-wave kernel: every wave runs alone) -> gpurun_out/wave_lives.npy [steps, waves]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
dev = torch.device("cuda", 0)
n = 8192
env = JitterbugVecEnv(n, "move_to_pose", seed=0, flags=32, stream=torch.cuda.current_stream(dev).cuda_stream)
env.randomise_models(seed=1000, return_params=False)
env.reset_device()
g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((400, n), generator=g, device=dev) * 2 - 1
obs = torch.zeros((n, env.obs_dim), device=dev); rew = torch.zeros((n,), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
out = []
for k in range(400):
    env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    if k >= 100:
        torch.cuda.synchronize(dev); out.append(env.wave_clocks() * 1e3)
np.save("gpurun_out/wave_lives.npy", np.stack(out).astype(np.float32))
print(np.stack(out).shape)
