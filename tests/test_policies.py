"""Heuristic policies against golden (obs -> action) vectors produced by running the reference's heuristic_policies.py
(tools/gen_golden_policies.py)."""
import collections
import json
import os

import numpy as np
import pytest

from jitterbug_amd import heuristic_policies as hp
from jitterbug_amd import model

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "policy_golden.json")))
GOLD, GOLD_KW = _G["cases"], _G["kwargs_cases"]


@pytest.mark.parametrize("task", model.TASKS)
def test_python_policies_match_reference(task):
    cases = [c for c in GOLD if c["task"] == task]
    obs = np.array([c["obs"] for c in cases])
    exp = np.array([c["action"] for c in cases])
    np.testing.assert_allclose(hp.policy_batch(task, obs), exp, rtol=0, atol=1e-15)
    assert len(set(np.round(exp, 6))) >= 2                        # the vectors exercise several branches (bang-bang: +-0.3 at least)
    # TimeStep call convention of the reference
    from jitterbug_amd.jitterbug import Jitterbug
    from jitterbug_amd.specs import StepType, TimeStep
    t = Jitterbug(task=task)
    for c in cases[:10]:
        ts = TimeStep(StepType.MID, 0.0, 1.0, t.split_observation(np.array(c["obs"])))
        assert hp.POLICIES[task](ts) == pytest.approx(c["action"], abs=1e-15)


def test_keyword_arguments_match_reference():
    """kick_angle / speed / orientation, angle_threshold, angle_to_target (reference heuristic_policies.py:6, 28, 64, 81, 98, 120):
    golden outputs of the reference functions called WITH those keyword arguments."""
    from jitterbug_amd.jitterbug import Jitterbug
    from jitterbug_amd.specs import StepType, TimeStep
    seen = collections.Counter()
    for c in GOLD_KW:
        ts = TimeStep(StepType.MID, 0.0, 1.0, Jitterbug(task=c["task"]).split_observation(np.array(c["obs"])))
        if c["fn"] == "optimal_orientation_to_move":
            a, orient = hp.optimal_orientation_to_move(ts, **c["kwargs"])
            assert a == pytest.approx(c["action"], abs=1e-15) and orient == c["orientation"]
        else:
            assert getattr(hp, c["fn"])(ts, **c["kwargs"]) == pytest.approx(c["action"], abs=1e-15), c
        seen[c["fn"]] += 1
    assert set(seen) == {"move_forward", "move_in_direction", "move_to_position", "move_to_pose", "face_direction", "optimal_orientation_to_move"}
    # batch form with the same keyword arguments
    for c in GOLD_KW:
        if c["fn"] in ("move_in_direction", "move_to_position", "move_to_pose"):
            assert hp.policy_batch(c["task"], np.array(c["obs"]), **c["kwargs"]) == pytest.approx(c["action"], abs=1e-15)


@pytest.mark.gpu
def test_device_policy_keyword_arguments():
    """jb_set_policy_params: the device policy with non-default kick_angle / speed / angle_threshold against the golden vectors."""
    from jitterbug_amd.vec_env import JitterbugVecEnv
    n_checked = 0
    for c in GOLD_KW:
        if c["fn"] == "move_forward" and c["kwargs"]["orientation"] in ("forward", "backward"):
            kw = dict(kick_angle=c["kwargs"]["kick_angle"], speed=c["kwargs"]["speed"])
        elif c["fn"] in ("move_in_direction", "move_to_position", "move_to_pose"):
            kw = dict(angle_threshold=c["kwargs"]["angle_threshold"])
        else:
            continue
        if n_checked >= 60:
            break
        env = JitterbugVecEnv(1, c["task"])
        env.set_policy_params(**kw)
        obs = np.array([c["obs"]], dtype=np.float32)
        exp32 = hp.policy_batch(c["task"], obs.astype(np.float64), **kw)
        assert abs(env.policy(obs)[0] - exp32[0]) < 2e-6
        env.close()
        n_checked += 1
    assert n_checked == 60


@pytest.mark.gpu
@pytest.mark.parametrize("task", model.TASKS)
def test_device_policies_match_reference(task):
    from jitterbug_amd.vec_env import JitterbugVecEnv
    cases = [c for c in GOLD if c["task"] == task]
    obs = np.array([c["obs"] for c in cases], dtype=np.float32)
    exp = np.array([c["action"] for c in cases])
    env = JitterbugVecEnv(len(cases), task)
    act = env.policy(obs)
    # fp32 observations: a branch may flip only if an observation sits within fp32 rounding of a threshold
    exp32 = hp.policy_batch(task, obs.astype(np.float64))
    np.testing.assert_allclose(act, exp32, rtol=0, atol=2e-6)
    assert (np.abs(act - exp) < 2e-6).mean() > 0.99
    env.close()


@pytest.mark.gpu
def test_policy_rollout_moves_the_robot():
    """evaluate_policy.py's loop (reference benchmarks/evaluate_policy.py:29-33) with the device policy: the heuristic
    drives move_from_origin to a positive return."""
    from jitterbug_amd.vec_env import JitterbugVecEnv
    env = JitterbugVecEnv(64, "move_from_origin", seed=0)
    obs = env.reset()
    total = np.zeros(64)
    for t in range(300):
        obs, r, d, _ = env.step(env.policy(obs))
        total += r
    assert total.mean() > 5.0
    env.close()


@pytest.mark.gpu
def test_evaluate_policy_single_env_and_batch_agree():
    """reference evaluate_policy loop on the dm_control-style env vs the chained device rollout of the same seeds."""
    from jitterbug_amd import suite
    from jitterbug_amd.evaluate_policy import evaluate_heuristic_batch, evaluate_policy
    env = suite.load("jitterbug", "move_in_direction", task_kwargs=dict(random=7, time_limit=0.4))
    single = evaluate_policy(env, hp.move_in_direction, num_repeats=1)
    assert single.shape == (1, 39) and np.isfinite(single).all()
    env.close()
    batch = evaluate_heuristic_batch("move_in_direction", num_repeats=32, seed=7, time_limit=0.4)
    assert batch.shape == (32, 39) and np.isfinite(batch).all() and (batch >= 0).all() and (batch <= 1).all()
    # env 0 of the batch has the same seed/stream as the single env; the first episode of the single env is its reset #1,
    # the batch rollout starts from reset #1 as well -> identical trajectories while fp32 policy decisions agree
    np.testing.assert_allclose(batch[0, :10], single[0, :10], rtol=1e-4, atol=1e-5)
