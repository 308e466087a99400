"""Writes tests/golden/oracle_trace.npz: the BASELINE configs[0]-shaped rollout (random policy, single envs) on the CPU
fp64 oracle.  This is a RESTATEMENT trace, not a MuJoCo trace (MuJoCo is not available here or
on the GPU box); a MuJoCo trace in the same format would be consumed by the same tests.

Per task: 2 envs, seed 0, 100 control steps, actions ~ U(-1, 1) from numpy default_rng(0).  Stored per step t:
qpos/qvel/target BEFORE the step, the action, and obs/reward AFTER it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jitterbug_amd import model
from oracle import oracle as O

T, N = 100, 2
out = {}
P = model.default_params()
for task in ("move_from_origin", "move_to_pose"):
    env = O.OracleEnv(N, task, P, seed=0)
    obs0 = env.reset()
    rng = np.random.default_rng(0)
    rec = {k: [] for k in ("qpos", "qvel", "target", "action", "obs", "reward")}
    for t in range(T):
        q, v, tg = env.get_state()
        a = rng.uniform(-1, 1, size=N)
        ob, rw, dn = env.step(a)
        for k, x in (("qpos", q), ("qvel", v), ("target", tg), ("action", a), ("obs", ob), ("reward", rw)):
            rec[k].append(np.array(x, dtype=np.float64))
    out[task + "/obs0"] = obs0
    for k in rec:
        out[task + "/" + k] = np.stack(rec[k])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_trace.npz"), **out)
print({k: v.shape for k, v in out.items()})
