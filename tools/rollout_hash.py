"""Prints a hash of a fixed rollout's observations / rewards (bit-level A/B of two library builds: JITTERBUG_HIP_LIB=... python tools/rollout_hash.py)."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
h = hashlib.sha256()
for task, n, act in (("move_from_origin", 4096, "uniform"), ("move_to_pose", 2048, "flat")):
    env = JitterbugVecEnv(n, task, seed=3)
    env.reset()
    rng = np.random.default_rng(0)
    for t in range(300):
        a = (rng.uniform(-1, 1, size=n) if act == "uniform" else np.ones(n)).astype(np.float32)
        ob, rw, dn, _ = env.step(a)
        h.update(ob.tobytes()); h.update(rw.tobytes())
    env.close()
print(os.environ.get("JITTERBUG_HIP_LIB", "default"), h.hexdigest()[:16])
