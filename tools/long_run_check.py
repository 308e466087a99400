import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
n = 4096
for name, law in (("const+1", lambda rng, t: np.ones(n)), ("bangbang", lambda rng, t: np.sign(rng.uniform(-1, 1, size=n))), ("uniform", lambda rng, t: rng.uniform(-1, 1, size=n))):
    env = JitterbugVecEnv(n, "move_to_pose", seed=7)
    env.reset(); rng = np.random.default_rng(0); t0 = time.time(); bad = 0
    for t in range(3000):
        ob, rw, dn, _ = env.step(law(rng, t).astype(np.float32))
        if not (np.isfinite(ob).all() and np.isfinite(rw).all()): bad += 1
    q, v, _ = env.get_state(); sc, ep, cap = env.counters()
    up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
    print("%-9s 3000 steps x %d envs: non-finite steps %d, |q|-1 max %.1e, z range [%.4f, %.4f], upright>0.9 %.3f, max |v| %.2f, cap hits total %.0f, %.1f s" % (name, n, bad, np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1).max(), q[:, 2].min(), q[:, 2].max(), (up > 0.9).mean(), np.abs(v[:, :3]).max(), cap.sum(), time.time() - t0), flush=True)
    env.close()
