import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd import model
from oracle import oracle as O
P = model.default_params()
n = 64
def mk(solver): return O.OracleEnv(n, "move_from_origin", P, seed=3, opts=O.default_opts(solver=solver))
A, B = mk(2), mk(1)      # 2: full-step active-set Newton (what the GPU does), 1: Newton with exact line search
A.reset(); B.reset()
for phase, T, law in (("uniform actions", 150, lambda rng: rng.uniform(-1, 1, size=n)), ("motor flat out", 500, lambda rng: np.ones(n))):
    rng = np.random.default_rng(0)
    it = {"A": [], "B": []}
    t0 = time.time()
    for t in range(T):
        a = law(rng)
        q, v, tg = A.get_state()
        B.set_state(q, v, tg)          # teacher-force B onto A's trajectory (its warm start stays its own)
        A.step(a, auto_reset=False); B.step(a, auto_reset=False)
        sa1, sb1 = A.stats(), B.stats()
        if t >= T - 100:
            it["A"].append(sa1.sweeps_total / max(1, sa1.nsolve)); it["B"].append(sb1.sweeps_total / max(1, sb1.nsolve))
    q, v, _ = A.get_state(); up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
    print("%s: upright>0.9 %.2f | Newton iterations per solve (mean over the last 100 steps): full-step %.3f, line search %.3f | sweeps_max full-step %d line-search %d | %.0f s" % (phase, (up > 0.9).mean(), np.mean(it["A"]), np.mean(it["B"]), A.stats().sweeps_max, B.stats().sweeps_max, time.time() - t0), flush=True)
