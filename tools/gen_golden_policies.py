"""Generates tests/golden/policy_golden.json by RUNNING the reference's heuristic policies (numpy only) on random
observation vectors.  reference: jitterbug_dmc/heuristic_policies.py:6-136, loaded standalone (it imports only numpy).
Cases without "kwargs" use the reference defaults; cases with "kwargs" pass the reference's keyword arguments
(kick_angle / speed / orientation of move_forward :28, angle_threshold of :64/:81/:98, angle_to_target of :6/:120).
    python tools/gen_golden_policies.py
"""
import collections
import importlib.util
import json
import os

import numpy as np

spec = importlib.util.spec_from_file_location("ref_heuristic_policies", "/root/reference/jitterbug_dmc/heuristic_policies.py")
hp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hp)

LAYOUT = {
    "move_from_origin": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1)],
    "face_direction": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("angle_to_target", 1)],
    "move_in_direction": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("angle_to_target", 1), ("speed_in_target_frame", 3)],
    "move_to_position": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("target_in_jitterbug_frame", 3)],
    "move_to_pose": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("target_in_jitterbug_frame", 3), ("angle_to_target", 1)],
}
TS = collections.namedtuple("TS", ["observation"])


def as_dict(task, v):
    obs, k = collections.OrderedDict(), 0
    for name, w in LAYOUT[task]:
        obs[name] = v[k:k + w].copy()
        k += w
    return obs


rng = np.random.default_rng(0)
cases = []
for task, layout in LAYOUT.items():
    D = sum(w for _, w in layout)
    for i in range(120):
        v = rng.uniform(-1, 1, size=D)
        if i % 5 == 0 and task in ("move_to_position", "move_to_pose"):
            v[15:17] *= 0.004                      # near the target: exercises the distance branch of move_to_pose
        if i % 7 == 0:
            v[13] = rng.choice([-0.9, -0.3, 0.0, 0.3, 0.9])
        a = getattr(hp, task)(TS(as_dict(task, v)))
        cases.append(dict(task=task, obs=v.tolist(), action=float(np.asarray(a).reshape(-1)[0])))

# keyword-argument cases (same generator stream, appended so the default cases above keep their values)
kw_cases = []
for i in range(60):
    v = rng.uniform(-1, 1, size=15)
    kw = dict(kick_angle=float(rng.uniform(0.1, 1.2)), speed=float(rng.uniform(0.05, 1.0)), orientation=str(rng.choice(["forward", "backward", "left", "right"])))
    kw_cases.append(dict(fn="move_forward", task="move_from_origin", obs=v.tolist(), kwargs=kw, action=float(hp.move_forward(TS(as_dict("move_from_origin", v)), **kw))))
for task in ("move_in_direction", "move_to_position", "move_to_pose"):
    D = sum(w for _, w in LAYOUT[task])
    for i in range(60):
        v = rng.uniform(-1, 1, size=D)
        kw = dict(angle_threshold=float(rng.uniform(0.02, 1.5)))
        a = getattr(hp, task)(TS(as_dict(task, v)), **kw)
        kw_cases.append(dict(fn=task, task=task, obs=v.tolist(), kwargs=kw, action=float(np.asarray(a).reshape(-1)[0])))
for i in range(40):
    v = rng.uniform(-1, 1, size=16)
    kw = dict(angle_to_target=float(rng.uniform(-np.pi, np.pi)))
    kw_cases.append(dict(fn="face_direction", task="face_direction", obs=v.tolist(), kwargs=kw, action=float(hp.face_direction(TS(as_dict("face_direction", v)), **kw))))
    a2, orient = hp.optimal_orientation_to_move(None, angle_to_target=kw["angle_to_target"])
    kw_cases.append(dict(fn="optimal_orientation_to_move", task="face_direction", obs=v.tolist(), kwargs=kw, action=float(a2), orientation=orient))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "policy_golden.json")
json.dump(dict(generator="tools/gen_golden_policies.py", source="reference heuristic_policies.py executed here", cases=cases, kwargs_cases=kw_cases), open(out, "w"))
print("wrote", out, len(cases), len(kw_cases))
