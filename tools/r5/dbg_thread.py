import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd import model
from jitterbug_amd.vec_env import JitterbugVecEnv
from oracle import oracle as O
from tests.test_thread_contact import _touching_models
models = _touching_models(4, seed=11)
for flags in (0, 8):
    P = np.stack([models[i % 4][0] for i in range(8)])
    g = JitterbugVecEnv(8, "move_to_pose", seed=4, auto_reset=False, params=P, flags=flags)
    o = O.OracleEnv(8, "move_to_pose", P, seed=4, per_env_model=True)
    g.reset(); o.reset()
    print("flags", flags, "variant", g.kernel_variant, "epw", g.envs_per_wave)
    for t in range(3):
        a = np.full(8, 0.5)
        q, v, tg = o.get_state()
        g.set_state(q, v, tg)
        og, _, _, _ = g.step(a)
        oo, _, _ = o.step(a, auto_reset=False)
        q1, v1, _ = o.get_state()
        qg, vg, _ = g.get_state()
        print(" step", t, "max obs err per env", np.round(np.abs(og - oo).max(axis=1), 6), "motor rate gpu/oracle", np.round(vg[:4, 14], 3), np.round(v1[:4, 14], 3))
    g.close()
