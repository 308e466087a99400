"""What tools/collect_rollout_profile.sh runs under rocprofv3: BASELINE configs[2] (4096 envs, move_from_origin, uniform action tape, seed 0),
one reset and ONE fused launch of K control steps (jb_step_many_device) - a whole episode for K = 1000.   python3 rollout_profile_target.py [K] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
env = JitterbugVecEnv(n, "move_from_origin", seed=0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
rew = torch.zeros((K, n), device=dev)
env.reset_device()
env.step_many_device(K, tape.data_ptr(), rewards_ptr=rew.data_ptr())
env.synchronize()
wc = env.wave_clocks()
print("fused K=%d n=%d: mean wave %.3f ms/step, slowest %.3f ms/step, finite %s" % (K, n, 1e3 * wc.mean() / K, 1e3 * wc.max() / K, bool(torch.isfinite(rew).all())))
env.close()
