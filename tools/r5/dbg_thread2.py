import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd import model
from jitterbug_amd.vec_env import JitterbugVecEnv
from oracle import oracle as O
from tests.test_thread_contact import _touching_models
models = _touching_models(16, seed=11)
n = 64
P = np.stack([models[i % len(models)][0] for i in range(n)])
g = JitterbugVecEnv(n, "move_to_pose", seed=4, auto_reset=False, params=P)
o = O.OracleEnv(n, "move_to_pose", P, seed=4, per_env_model=True)
g.reset(); o.reset()
rng = np.random.default_rng(4)
saved = []
for t in range(150):
    a = rng.uniform(-1, 1, size=n)
    q, v, tg = o.get_state()
    g.set_state(q, v, tg)
    og, _, _, _ = g.step(a)
    oo, _, _ = o.step(a, auto_reset=False)
    mar = o.margins()
    err = np.abs(og.astype(np.float64) - oo)
    w = err <= 1e-4 * np.abs(oo) + 1e-6
    for i in np.nonzero((~w).any(axis=1) & (mar >= 3e-8))[0]:
        geoms = set()
        for sub in range(0, 50, 5):
            qs, vs = O.step_physics(P[i], q[i], v[i], a[i], sub) if sub else (q[i], v[i])
            d = O.forward_debug(P[i], qs, vs, a[i])
            geoms |= set(int(x) for x in d["con_geom"][:d["ncon"]])
        print("step", t, "env", i, "bad entries", int((~w[i]).sum()), "max err %.2e" % err[i].max(), "margin %.2e" % mar[i], "contact geoms over the step", sorted(geoms))
        if len(saved) < 6: saved.append(np.concatenate([[t, i, a[i]], q[i], v[i], tg[i]]))
np.save("gpurun_out/r5_thread_bad.npy", np.array(saved))
print("resolved", g.solver_stats(), "cap", g.counters()[2].sum())
