"""Writes tests/golden/oracle_trace.npz: the BASELINE configs[0]-shaped rollout (random policy, single envs) on the CPU
fp64 oracle.  This is a RESTATEMENT trace, not a MuJoCo trace (MuJoCo is not available here or
on the GPU box); a MuJoCo trace in the same format would be consumed by the same tests.

move_from_origin: ONE env, 1000 control steps (a whole episode: exactly configs[0]); move_to_pose: 2 envs, 100 steps.  Seed 0,
actions ~ U(-1, 1) from numpy default_rng(0).  Stored per step t: qpos/qvel/target BEFORE the step, the action, obs/reward AFTER it,
and the oracle's contact-switch margin of the step (jbo_stats.margin_min, see tests/test_gpu_parity.py MARGIN_TOL)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jitterbug_amd import model
from oracle import oracle as O

out = {}
P = model.default_params()
for task, T, N in (("move_from_origin", 1000, 1), ("move_to_pose", 100, 2)):
    env = O.OracleEnv(N, task, P, seed=0, step_limit=10 ** 9)
    obs0 = env.reset()
    rng = np.random.default_rng(0)
    rec = {k: [] for k in ("qpos", "qvel", "target", "action", "obs", "reward", "margin")}
    for t in range(T):
        q, v, tg = env.get_state()
        a = rng.uniform(-1, 1, size=N)
        ob, rw, dn = env.step(a)
        for k, x in (("qpos", q), ("qvel", v), ("target", tg), ("action", a), ("obs", ob), ("reward", rw), ("margin", env.margins())):
            rec[k].append(np.array(x, dtype=np.float64))
    out[task + "/obs0"] = obs0
    for k in rec:
        out[task + "/" + k] = np.stack(rec[k])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_trace.npz"), **out)
print({k: v.shape for k, v in out.items()})
