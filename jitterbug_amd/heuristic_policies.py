"""Heuristic bang-bang policies for the five tasks (reference: jitterbug_dmc/heuristic_policies.py:6-136).

Two forms:
  * the reference's call convention — ``move_to_pose(ts)`` etc. on a TimeStep with the observation dict;
  * batch form ``policy_batch(task, obs[N, D]) -> actions[N]`` on flat observation rows, which is also what the
    device kernel behind ``jb_policy_device`` computes (no host round trip between observation and action).
The policies read the NORMALISED observations, exactly like the reference (e.g. ``angle_to_target`` is angle/pi and is
compared against thresholds in radians; motor_position is angle/pi compared against the kick angle in radians) — the
quirk is kept because it defines the behaviour the reference's evaluate_policy.py measures."""
import numpy as np

from . import model

KICK_ANGLE = np.deg2rad(45)
SPEED = 0.3
ANGLE_THRESHOLD = np.deg2rad(20)


def _face(angle):                                              # reference :6-25
    return 0.9 * np.clip(3 * angle / np.pi, -1, 1)


def _forward(motor_angle, motor_vel, offset):                  # reference :28-56
    bang = np.where(motor_vel > 0, SPEED, -SPEED)
    return np.where(motor_angle < offset - KICK_ANGLE, SPEED, np.where(motor_angle > offset + KICK_ANGLE, -SPEED, bang))


def _optimal_orientation(angle):                               # reference :120-136  -> (angle', offset of move_forward)
    q = np.pi / 4
    left = (angle > q) & (angle <= np.pi)
    right = (angle >= -np.pi) & (angle < -q)
    folded = np.abs(np.abs(angle) - np.pi / 2)
    a = np.where(left, folded, np.where(right, -folded, angle))
    off = np.where(left, np.pi / 2, np.where(right, -np.pi / 2, 0.0))
    return a, off


def policy_batch(task, obs):
    obs = np.asarray(obs, dtype=np.float64)
    squeeze = obs.ndim == 1
    obs = np.atleast_2d(obs)
    ma, mv = obs[:, 13], obs[:, 14]
    if task == "move_from_origin":                             # :59-61
        act = _forward(ma, mv, 0.0)
    elif task == "face_direction":
        act = _face(obs[:, 15])
    elif task in ("move_in_direction", "move_to_position"):   # :64-95
        ang = obs[:, 15] if task == "move_in_direction" else np.arctan2(obs[:, 15], -obs[:, 16])
        a, off = _optimal_orientation(ang)
        act = np.where(np.abs(a) > ANGLE_THRESHOLD, _face(a), _forward(ma, mv, off))
    elif task == "move_to_pose":                               # :98-118
        dx, dy = obs[:, 15], obs[:, 16]
        ang = np.arctan2(dx, -dy)
        near = np.hypot(dx, dy) <= 0.01
        act = np.where(np.abs(ang) > ANGLE_THRESHOLD, _face(ang), np.where(near, _face(obs[:, 18]), _forward(ma, mv, 0.0)))
    else:
        raise ValueError("Invalid task {}".format(task))
    return act[0] if squeeze else act


def _flat(ts, task):
    o = ts.observation
    if "observations" in o:
        return np.asarray(o["observations"], dtype=np.float64)
    return np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in o.values()])


def _make(task):
    def policy(ts):
        return float(policy_batch(task, _flat(ts, task)))
    policy.__name__ = task
    policy.__doc__ = "Heuristic policy for %s (reference heuristic_policies.py)" % task
    return policy


move_from_origin = _make("move_from_origin")
face_direction = _make("face_direction")
move_in_direction = _make("move_in_direction")
move_to_position = _make("move_to_position")
move_to_pose = _make("move_to_pose")
POLICIES = {t: globals()[t] for t in model.TASKS}
