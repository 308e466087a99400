"""JitterbugVecEnv.rollout(100) ten times (jb_step_many, host buffers): the device staging belongs to the handle and only grows, so a HIP trace
shows allocations in the first call only.   rocprofv3 --hip-trace ... -- python3 tools/r5/rollout10.py ; tools/r5/rollout_alloc_count.py <dir>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
env = JitterbugVecEnv(4096, "move_from_origin", seed=0)
env.reset()
rng = np.random.default_rng(0)
tape = rng.uniform(-1, 1, size=(100, 4096)).astype(np.float32)
for i in range(10):
    t0 = time.time()
    ob, rw, dn = env.rollout(100, tape)
    rw2, _ = env.rollout_policy(50)
    print("call %d: %.1f ms, finite %s" % (i, 1e3 * (time.time() - t0), bool(np.isfinite(ob).all() and np.isfinite(rw2).all())), flush=True)
    with open(os.environ.get("JB_MARK_FILE", "/tmp/jb_marks.txt"), "a") as f:
        f.write("%d %d\n" % (i, time.time_ns()))
env.close()
