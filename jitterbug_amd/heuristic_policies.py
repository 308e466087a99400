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

KICK_ANGLE = np.deg2rad(45)          # reference :28
SPEED = 0.3                          # reference :28
ANGLE_THRESHOLD = np.deg2rad(20)     # reference :64, :81, :98
_OFFSETS = {"forward": 0.0, "backward": 0.0, "left": np.pi / 2, "right": -np.pi / 2, "None": 0.0}     # reference :43-47


def _face(angle):                                              # reference :6-25
    return 0.9 * np.clip(3 * angle / np.pi, -1, 1)


def _forward(motor_angle, motor_vel, offset, kick_angle=KICK_ANGLE, speed=SPEED):      # reference :28-56
    bang = np.where(motor_vel > 0, speed, -speed)
    return np.where(motor_angle < offset - kick_angle, speed, np.where(motor_angle > offset + kick_angle, -speed, bang))


def _optimal_orientation(angle):                               # reference :120-136  -> (angle', offset of move_forward)
    q = np.pi / 4
    left = (angle > q) & (angle <= np.pi)
    right = (angle >= -np.pi) & (angle < -q)
    folded = np.abs(np.abs(angle) - np.pi / 2)
    a = np.where(left, folded, np.where(right, -folded, angle))
    off = np.where(left, np.pi / 2, np.where(right, -np.pi / 2, 0.0))
    return a, off


def policy_batch(task, obs, *, kick_angle=KICK_ANGLE, speed=SPEED, angle_threshold=ANGLE_THRESHOLD):
    """Flat observation rows [N, D] (or one row) -> actions; the keyword arguments are the reference policies' own."""
    obs = np.asarray(obs, dtype=np.float64)
    squeeze = obs.ndim == 1
    obs = np.atleast_2d(obs)
    ma, mv = obs[:, 13], obs[:, 14]
    fwd = lambda off: _forward(ma, mv, off, kick_angle, speed)
    if task == "move_from_origin":                             # :59-61
        act = fwd(0.0)
    elif task == "face_direction":
        act = _face(obs[:, 15])
    elif task in ("move_in_direction", "move_to_position"):   # :64-95
        ang = obs[:, 15] if task == "move_in_direction" else np.arctan2(obs[:, 15], -obs[:, 16])
        a, off = _optimal_orientation(ang)
        act = np.where(np.abs(a) > angle_threshold, _face(a), fwd(off))
    elif task == "move_to_pose":                               # :98-118
        dx, dy = obs[:, 15], obs[:, 16]
        ang = np.arctan2(dx, -dy)
        near = np.hypot(dx, dy) <= 0.01
        act = np.where(np.abs(ang) > angle_threshold, _face(ang), np.where(near, _face(obs[:, 18]), fwd(0.0)))
    else:
        raise ValueError("Invalid task {}".format(task))
    return act[0] if squeeze else act


def _flat(ts):
    o = ts.observation
    if "observations" in o:
        return np.asarray(o["observations"], dtype=np.float64)
    return np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in o.values()])


def _entry(ts, name, index=0):
    """ts.observation[name][index] from the dict form, or from the flat 'observations' vector (all tasks share the first 15)."""
    o = ts.observation
    if name in o:
        return float(np.asarray(o[name], dtype=np.float64).reshape(-1)[index])
    return float(np.asarray(o["observations"], dtype=np.float64)[{"motor_position": 13, "motor_velocity": 14}[name] + index])


# ---- the reference's call convention, keyword arguments included (reference heuristic_policies.py:6-136)
def face_direction(ts, *, angle_to_target=None):                                      # :6
    if angle_to_target is None:
        angle_to_target = ts.observation["angle_to_target"]        # dict observations, like the reference
    return float(_face(float(np.asarray(angle_to_target).reshape(-1)[0])))


def move_forward(ts, *, kick_angle=KICK_ANGLE, speed=SPEED, orientation="forward"):   # :28
    return float(_forward(_entry(ts, "motor_position"), _entry(ts, "motor_velocity"), _OFFSETS.get(orientation, 0.0), kick_angle, speed))


def move_from_origin(ts):                                                             # :59
    return move_forward(ts)


def optimal_orientation_to_move(ts, *, angle_to_target):                              # :120
    a = float(np.asarray(angle_to_target).reshape(-1)[0])
    orientation = "None"
    if -np.pi / 4 <= a <= np.pi / 4:
        orientation = "forward"
    elif np.pi / 4 < a <= np.pi:
        orientation, a = "left", abs(abs(a) - np.pi / 2)
    elif -np.pi <= a < -np.pi / 4:
        orientation, a = "right", -abs(abs(a) - np.pi / 2)
    return [a, orientation]


def move_in_direction(ts, *, angle_threshold=ANGLE_THRESHOLD):                        # :64
    return float(policy_batch("move_in_direction", _flat(ts), angle_threshold=angle_threshold))


def move_to_position(ts, *, angle_threshold=ANGLE_THRESHOLD):                         # :81
    return float(policy_batch("move_to_position", _flat(ts), angle_threshold=angle_threshold))


def move_to_pose(ts, *, angle_threshold=ANGLE_THRESHOLD):                             # :98
    return float(policy_batch("move_to_pose", _flat(ts), angle_threshold=angle_threshold))


POLICIES = {t: globals()[t] for t in model.TASKS}
