import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from jitterbug_amd.vec_env import JitterbugVecEnv
dev = torch.device("cuda", 0)
for const in (False, True):
    n = 4096
    env = JitterbugVecEnv(n, "move_from_origin", seed=1, max_newton=int(os.environ.get("JB_MAX_NEWTON", "12")))
    g = torch.Generator(device=dev); g.manual_seed(5)
    tape = torch.rand((1000, n), generator=g, device=dev) * 2 - 1
    if const: tape.fill_(1.0)
    env.reset_device()
    for e in range(6):
        env.step_many_device(1000, tape.data_ptr())
    env.synchronize()
    sc, ep, cap = env.counters()
    wc = env.wave_clocks()
    print("const1" if const else "uniform", "cap hits", float(cap.sum()), "mean wave ms/step %.4f" % (wc.mean()))
    env.close()
