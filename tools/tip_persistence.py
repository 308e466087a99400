"""How long does a robot that has tipped over stay tipped?  (Round 4: decides what a fused K-step rollout can gain - a wave that
owns its envs for K steps is bounded by its own SUM over the steps, so persistent slow envs bound the rollout, transient ones do not.)

    python tools/tip_persistence.py [n_envs] [steps] [actions: uniform|const1] > gpurun_out/tip_persistence.txt

Uprightness Rzz = 1 - 2 (qx^2 + qy^2) of every env at every control step of one episode (BASELINE configs[2], seed 0).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mode = sys.argv[3] if len(sys.argv) > 3 else "uniform"
env = JitterbugVecEnv(n, "move_from_origin", seed=0, auto_reset=False)
env.reset()
rng = np.random.default_rng(1234)
rzz = np.zeros((steps, n), dtype=np.float32)
pz = np.zeros((steps, n), dtype=np.float32)
for t in range(steps):
    a = np.ones(n, np.float32) if mode == "const1" else rng.uniform(-1, 1, size=n).astype(np.float32)
    obs, _, _, _ = env.step(a)
    rzz[t] = 1 - 2 * (obs[:, 4] ** 2 + obs[:, 5] ** 2)
    pz[t] = (obs[:, 2] + 1) / 20
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tip_persistence_%s.npz" % mode), rzz=rzz.astype(np.float16), pz=pz.astype(np.float16))
for thr in (0.95, 0.9, 0.8, 0.5):
    tip = rzz < thr
    print("Rzz < %.2f: tipped envs at steps 100/200/400/600/800/999: %s;  env-steps tipped %.3f %%" % (thr, [int(tip[min(s, steps - 1)].sum()) for s in (100, 200, 400, 600, 800, 999)], 100 * tip.mean()))
    # persistence: of the envs tipped at step t, the share still tipped d steps later
    for d in (10, 50, 100, 300):
        num = den = 0
        for t in range(0, steps - d, 10):
            den += int(tip[t].sum()); num += int((tip[t] & tip[t + d]).sum())
        print("    still tipped %3d steps later: %.3f  (of %d samples)" % (d, num / max(den, 1), den))
    per_env = tip.sum(0)
    print("    envs ever tipped: %d; tipped-steps per such env: mean %.0f, median %.0f, max %d;  envs tipped > 50 %% of the episode: %d" % (
        int((per_env > 0).sum()), per_env[per_env > 0].mean() if (per_env > 0).any() else 0, np.median(per_env[per_env > 0]) if (per_env > 0).any() else 0, int(per_env.max()), int((per_env > steps / 2).sum())))
