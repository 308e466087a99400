"""Where does a host-buffer step spend its time?  jb_step against jb_step_async / jb_step_wait, call by call (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = JitterbugVecEnv(n, "move_from_origin", seed=0)
env.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, size=(64, n)).astype(np.float32)
for i in range(30):
    env.step(acts[i % 64])
K = 300
t0 = time.perf_counter()
for i in range(K):
    env.step(acts[i % 64])
sync_ms = (time.perf_counter() - t0) * 1e3 / K
ta = tw = 0.0
for i in range(30):
    env.step_async(acts[i % 64]); env.step_wait()
t0 = time.perf_counter()
for i in range(K):
    t1 = time.perf_counter()
    env.step_async(acts[i % 64])
    t2 = time.perf_counter()
    env.step_wait(copy=False)
    t3 = time.perf_counter()
    ta += t2 - t1; tw += t3 - t2
both_ms = (time.perf_counter() - t0) * 1e3 / K
t0 = time.perf_counter()
for i in range(K):
    env.step_async(acts[i % 64]); env.step_wait()
copy_ms = (time.perf_counter() - t0) * 1e3 / K
print("N = %d: jb_step %.4f ms | step_async call %.4f ms + step_wait(copy=False) %.4f ms = %.4f ms per step | with copies %.4f ms" % (n, sync_ms, ta * 1e3 / K, tw * 1e3 / K, both_ms, copy_ms))
