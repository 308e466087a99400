"""Full-episode returns under simple open-loop action laws, per task (GPU simulator).  Compare with the START of the
training curves of the reference's fig-rl-perf.ipynb (real MuJoCo, nearly untrained policies)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd import model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
laws = {
    "uniform(-1,1)": lambda rng, t, n: rng.uniform(-1, 1, size=n),
    "gauss(0,0.3)": lambda rng, t, n: np.clip(rng.normal(0, 0.3, size=n), -1, 1),
    "const +1": lambda rng, t, n: np.ones(n),
    "const -1": lambda rng, t, n: -np.ones(n),
    "const 0.3": lambda rng, t, n: 0.3 * np.ones(n),
}
for task in model.TASKS:
    for name, law in laws.items():
        env = JitterbugVecEnv(n, task, seed=0, auto_reset=False)
        env.reset()
        rng = np.random.default_rng(1)
        ret = np.zeros(n)
        for t in range(999):
            ob, rw, dn, _ = env.step(law(rng, t, n).astype(np.float32))
            ret += rw
        q, v, tg = env.get_state()
        d = np.hypot(q[:, 0], q[:, 1])
        up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
        print("%-18s %-14s return mean %.0f median %.0f p10 %.0f p90 %.0f | final dist from origin mean %.3f m, upright>0.9: %.3f, motor rate mean %.0f" %
              (task, name, ret.mean(), np.median(ret), np.quantile(ret, .1), np.quantile(ret, .9), d.mean(), (up > 0.9).mean(), np.abs(v[:, 14]).mean()), flush=True)
        env.close()
        if task not in ("move_from_origin", "move_in_direction") and name != "uniform(-1,1)": break
