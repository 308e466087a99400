"""Behavioural probe (no parity claim): the reference's heuristic policies (heuristic_policies.py:6-136) were
written for angles in RADIANS, but at HEAD they are fed the normalised observations (angle / pi).  This runs them
closed loop on the GPU simulator both ways - HEAD semantics (normalised inputs) and the semantics they were written for
(de-normalised inputs) - and reports full-episode returns per task against the paper's "solves each task" (return >~ 900,
manuscript/ICRA2020/root.tex:279,392), plus how far the move_from_origin policy travels (the reference's heat-map shows ~2 m for a
trained policy).  python tools/heuristic_radians.py [n_envs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd import heuristic_policies as hp, model
from jitterbug_amd.vec_env import JitterbugVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def denormalise(task, obs):
    o = np.array(obs, dtype=np.float64)
    o[:, 13] *= np.pi; o[:, 14] *= 180.0                       # motor angle (rad), motor rate (rad/s): jitterbug.py:345-346
    if task in ("face_direction", "move_in_direction"):
        o[:, 15] *= np.pi                                       # angle_to_target
    if task in ("move_to_position", "move_to_pose"):
        o[:, 15] *= 3.0; o[:, 16] *= 3.0; o[:, 17] /= 10.0      # target_in_jitterbug_frame (m)
    if task == "move_to_pose":
        o[:, 18] *= np.pi
    return o


for task in model.TASKS:
    for label, prep in (("HEAD (normalised inputs)", lambda t, o: o), ("radians (de-normalised)", denormalise)):
        env = JitterbugVecEnv(n, task, seed=0, auto_reset=False)
        ob = env.reset()
        ret = np.zeros(n); far = np.zeros(n)
        for t in range(999):
            a = hp.policy_batch(task, prep(task, ob) if prep is denormalise else ob)
            ob, rw, dn, _ = env.step(a.astype(np.float32))
            ret += rw
            if t % 50 == 49:
                q, _, _ = env.get_state()
                far = np.maximum(far, np.hypot(q[:, 0], q[:, 1]))
        q, v, tg = env.get_state()
        d = np.hypot(q[:, 0], q[:, 1])
        up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
        print("%-18s %-26s return mean %4.0f median %4.0f p10 %4.0f p90 %4.0f  solved(>=900) %.2f | final distance from origin mean %.3f m max %.3f m (farthest on the way %.3f mean), upright>0.9 %.3f"
              % (task, label, ret.mean(), np.median(ret), np.quantile(ret, .1), np.quantile(ret, .9), (ret >= 900).mean(), d.mean(), d.max(), far.mean(), (up > 0.9).mean()), flush=True)
        env.close()
