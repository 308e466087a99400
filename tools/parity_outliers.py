"""Which entries of which env-steps are outside the parity tolerance on well-conditioned steps?  (diagnostic)
python tools/parity_outliers.py task n steps seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd import model
from jitterbug_amd.vec_env import JitterbugVecEnv
from oracle import oracle as O
task, n, steps, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
flat_out = len(sys.argv) > 5 and sys.argv[5] == "flat"
skip = int(sys.argv[6]) if len(sys.argv) > 6 else 0
P = model.default_params()
g = JitterbugVecEnv(n, task, seed=seed, auto_reset=False, max_newton=int(os.environ.get("JB_MAX_NEWTON", "12")))
o = O.OracleEnv(n, task, P, seed=seed)
g.reset(); o.reset()
rng = np.random.default_rng(seed)
cnt = np.zeros(model.OBS_DIM[task], dtype=int)
shown = 0
for t in range(-skip, steps):
    a = np.ones(n) if flat_out else rng.uniform(-1, 1, size=n)
    if t < 0:
        o.step(a, auto_reset=False)
        continue
    q, v, tg = o.get_state()
    g.set_state(q, v, tg)
    cap0 = g.counters()[2].copy()
    og, rg, dg, _ = g.step(a)
    capd = g.counters()[2] - cap0
    oo, ro, do = o.step(a, auto_reset=False)
    well = o.margins() >= 3e-8
    og = og.astype(np.float64)
    err = np.abs(og - oo)
    bad = (err > 1e-4 * np.abs(oo) + 1e-6) & well[:, None]
    cnt += bad.sum(0)
    stat = globals().setdefault("stat", dict(bad_steps=0, bad_with_cap=0, cap_steps=0, cap_steps_bad=0))
    bs = bad.any(1)
    stat["bad_steps"] += int(bs.sum()); stat["bad_with_cap"] += int((bs & (capd > 0)).sum()); stat["cap_steps"] += int((capd > 0).sum()); stat["cap_steps_bad"] += int(((capd > 0) & (err > 1e-4 * np.abs(oo) + 1e-6).any(1)).sum())
    for i, j in zip(*np.nonzero(bad)):
        if shown < 25:
            shown += 1
            Rzz = 1 - 2 * (q[i, 4] ** 2 + q[i, 5] ** 2)
            R00 = 1 - 2 * (q[i, 5] ** 2 + q[i, 6] ** 2); R10 = 2 * (q[i, 4] * q[i, 5] + q[i, 3] * q[i, 6])
            print("step %4d env %3d obs[%2d]: gpu %+.6e oracle %+.6e err %.2e | margin %.2e | Rzz %.2f heading lever %.3f | all errs of this env-step: %s" % (
                t, i, j, og[i, j], oo[i, j], err[i, j], o.margins()[i], Rzz, np.hypot(R00, R10), " ".join("%.0e" % e for e in err[i])))
print("bad entries per observation index:", cnt.tolist())
print("env-steps with a well-conditioned entry outside the tolerance: %(bad_steps)d, of which the Newton cap was hit in that step: %(bad_with_cap)d; env-steps with a cap hit: %(cap_steps)d, of which any entry outside the tolerance: %(cap_steps_bad)d" % stat)
