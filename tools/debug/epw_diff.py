"""Where do two envs-per-wave variants of the step kernel first disagree?  (diagnostic)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, size=(steps, n)).astype(np.float32)
if len(sys.argv) > 3 and sys.argv[3] == "const":
    acts[:] = 1.0          # motor flat out: most robots tip over and lie on a leg (spread sweeps)
res = {}
for epw in (1, 2, 4, 8):
    e = JitterbugVecEnv(n, "move_from_origin", seed=5, envs_per_wave=epw)
    ob = [e.reset()]
    for a in acts:
        ob.append(e.step(a)[0])
    ob = ob[-50:]
    res[epw] = np.stack(ob)
    e.close()
for epw in (1, 2, 8):
    d = res[epw] != res[4]
    print("epw", epw, "vs 4: differing entries", int(d.sum()), "envs", int(d.any(axis=(0, 2)).sum()))
    if d.any():
        t = int(np.argmax(d.any(axis=(1, 2))))
        envs = np.nonzero(d[t].any(axis=1))[0]
        print("  first step", t, "envs", envs[:10], "max abs diff", float(np.abs(res[epw][t] - res[4][t]).max()))
        print("  max abs diff overall", float(np.abs(res[epw] - res[4]).max()))
