"""The committed restatement trace (tests/golden/oracle_trace.npz, written by tools/gen_golden_trace.py; SURVEY.md §8d
config 1).  It is NOT a MuJoCo trace: MuJoCo cannot run here or on the GPU box ('parity unpinned').  What it pins:
  * CPU: the oracle still reproduces it (teacher-forced per step, 1e-9) - a change of the oracle's arithmetic is caught;
  * GPU: the HIP path, teacher-forced on the trace's states through the C ABI, matches its observations / rewards to the
    fp32 tolerance of the parity tests: every entry of every well-conditioned step within 1e-4 rel + 1e-6 abs (conditioning:
    tests/test_gpu_parity.py MARGIN_TOL), >= 99.9 % of all entries.
move_from_origin is BASELINE configs[0] in full: one env, random policy, 1000 control steps."""
import os
import numpy as np
import pytest

from jitterbug_amd import model

TRACE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_trace.npz")
TASKS = ("move_from_origin", "move_to_pose")


def _load(task):
    z = np.load(TRACE)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(task + "/")}


@pytest.mark.parametrize("task", TASKS)
def test_oracle_reproduces_trace(task):
    from oracle import oracle as O
    tr = _load(task)
    T, n = tr["action"].shape
    env = O.OracleEnv(n, task, model.default_params(), seed=0, step_limit=10 ** 9)
    obs0 = env.reset()
    np.testing.assert_allclose(obs0, tr["obs0"], rtol=0, atol=1e-12)          # reset stream (Philox) and reset observation
    q, v, tg = env.get_state()
    np.testing.assert_allclose(q, tr["qpos"][0], rtol=0, atol=1e-12)
    for t in range(T):
        env.set_state(tr["qpos"][t], tr["qvel"][t], tr["target"][t])
        ob, rw, dn = env.step(tr["action"][t], auto_reset=False)
        np.testing.assert_allclose(ob, tr["obs"][t], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(rw, tr["reward"][t], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(env.margins(), tr["margin"][t], rtol=1e-6, atol=1e-15)
    # physical sanity of the recorded rollout: the robot stays above the floor and the unit quaternion stays unit
    assert (tr["qpos"][:, :, 2] > 0.0).all()
    np.testing.assert_allclose(np.linalg.norm(tr["qpos"][:, :, 3:7], axis=-1), 1.0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("task", TASKS)
def test_hip_matches_trace(task):
    from jitterbug_amd.vec_env import JitterbugVecEnv
    tr = _load(task)
    T, n = tr["action"].shape
    g = JitterbugVecEnv(n, task, seed=0, auto_reset=False, time_limit=float("inf"))
    og = g.reset()
    assert np.abs(og - tr["obs0"]).max() < 1e-6
    ok = tot = well_bad = ill = 0
    worst = 0.0
    for t in range(T):
        g.set_state(tr["qpos"][t], tr["qvel"][t], tr["target"][t])
        ob, rw, dn, _ = g.step(tr["action"][t].astype(np.float32))
        d = np.abs(ob - tr["obs"][t])
        good = d <= 1e-4 * np.abs(tr["obs"][t]) + 1e-6
        ok += good.sum(); tot += good.size
        well = tr["margin"][t] >= 3e-8               # tests/test_gpu_parity.py MARGIN_TOL: the step is not within 30 nm of a contact switch
        well_bad += (~good[well]).sum(); ill += (~well).sum()
        if well.any():
            worst = max(worst, d[well].max())
            assert np.abs(rw - tr["reward"][t])[well].max() < 1e-5
    print("trace parity %s: %.5f of entries within tolerance, worst abs %.2e on well-conditioned steps, %d of %d env-steps ill-conditioned"
          % (task, ok / tot, worst, ill, T * n))
    assert well_bad == 0 and ok / tot >= 0.999 and ill <= 0.02 * T * n
    g.close()
