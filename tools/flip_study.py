"""CPU study of the teacher-forced parity outliers (no GPU): the kernel's simulator source compiled for the host in fp32
(tests/host_harness.cpp) against the fp64 oracle, env-step by env-step, together with the oracle's contact-switch margin
(jbo_stats.margin_min: the smallest |distance| of any contact candidate at a substep boundary).
    python tools/flip_study.py [n_envs] [steps] [task]
Prints the fraction of observation entries inside the north-star tolerance, overall and as a function of the margin."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jitterbug_amd import model  # noqa: E402
from oracle import oracle as O  # noqa: E402
import tests.build_harness as bh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
task = sys.argv[3] if len(sys.argv) > 3 else "move_from_origin"
lib = C.CDLL(bh.build())
dp = C.POINTER(C.c_double)
lib.jbh_step.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
P = model.default_params()


def hstep(q, v, u, f32=1):
    q, v, fail = q.copy(), v.copy(), np.zeros(1)
    rc = lib.jbh_step(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), float(u), 50, 1, 20, 1, f32, fail.ctypes.data_as(dp))
    assert rc == 0
    qn = q[3:7] / np.linalg.norm(q[3:7]); q[3:7] = qn
    return q, v


env = O.OracleEnv(n, task, P, seed=3)
env.reset()
rng = np.random.default_rng(3)
rows = []
for t in range(steps):
    a = rng.uniform(-1, 1, size=n)
    q0, v0, tg = env.get_state()
    oo, _, _ = env.step(a, auto_reset=False)
    mar = env.margins()
    for i in range(n):
        qf, vf = hstep(q0[i], v0[i], np.float32(a[i]))          # (the harness splits the fp64 state into hi + lo words like jb_set_state)
        of = O.observation(P, task, qf, vf, tg[i])
        err = np.abs(of - oo[i])
        ok = err <= 1e-4 * np.abs(oo[i]) + 1e-6
        rows.append((mar[i], ok.mean(), err.max()))
rows = np.array(rows)
print("env-steps %d; entries within tolerance %.5f; env-steps fully within %.5f; worst %.3g" % (len(rows), rows[:, 1].mean(), (rows[:, 1] == 1).mean(), rows[:, 2].max()))
for lo, hi in ((0, 1e-9), (1e-9, 1e-8), (1e-8, 3e-8), (3e-8, 1e-7), (1e-7, 3e-7), (3e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1)):
    m = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if m.any():
        print("margin [%.0e, %.0e): %6d env-steps (%.2f %%), fully within tolerance %.4f, worst error %.3g" % (lo, hi, m.sum(), 100 * m.mean(), (rows[m, 1] == 1).mean(), rows[m, 2].max()))
