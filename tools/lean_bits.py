import numpy as np, sys
sys.path.insert(0, '/root/repo')
from jitterbug_amd.vec_env import JitterbugVecEnv
for regime in ("uniform", "flat"):
    outs = []
    for flags in (0, 2):
        g = JitterbugVecEnv(1024, "move_to_pose", seed=3, flags=flags, envs_per_wave=4)
        ob = g.reset(); rng = np.random.default_rng(0); acc = []
        for t in range(300 if regime == "flat" else 120):
            a = np.ones(1024, np.float32) if regime == "flat" else rng.uniform(-1, 1, size=1024).astype(np.float32)
            ob, rw, dn, _ = g.step(a); 
            if t % 20 == 19: acc.append(ob.copy())
        q, v, _ = g.get_state()
        outs.append((np.stack(acc), q, v)); g.close()
    same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    d = max(np.abs(a - b).max() for a, b in zip(outs[0], outs[1]))
    print(regime, "lean == ordinary bitwise:", same, "max diff", d, "tipped", float(((1 - 2 * (outs[0][1][:, 4] ** 2 + outs[0][1][:, 5] ** 2)) < 0.5).mean()))
