"""Diagnostic: cost of per-env model tables (same nominal model replicated) vs the shared table vs randomised models."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd import model
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd.augmented_jitterbug import augmented_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
def run(mode):
    env = JitterbugVecEnv(n, "move_from_origin", seed=0)
    if mode == "replicated": env.set_model_params(np.tile(model.default_params(), (n, 1)))
    if mode == "augmented": env.set_model_params(augmented_params(n, seed=1000))
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(260, n)).astype(np.float32)
    for t in range(60): env.step(acts[t])
    t0 = time.perf_counter()
    for t in range(60, 260): env.step(acts[t])
    dt = time.perf_counter() - t0
    q, v, _ = env.get_state()
    up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
    print("%-11s %.0f env-steps/s (host-buffer API), upright<0.5: %.4f, mean z %.4f" % (mode, n * 200 / dt, (up < 0.5).mean(), q[:, 2].mean()), flush=True)
    env.close()
for mode in ("shared", "replicated", "augmented"): run(mode)
