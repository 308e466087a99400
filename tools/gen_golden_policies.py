"""Generates tests/golden/policy_golden.json by RUNNING the reference's heuristic policies (numpy only) on random
observation vectors.  reference: jitterbug_dmc/heuristic_policies.py:6-136, loaded standalone (it imports only numpy).
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
        obs, k = collections.OrderedDict(), 0
        for name, w in layout:
            obs[name] = v[k:k + w].copy()
            k += w
        a = getattr(hp, task)(TS(obs))
        cases.append(dict(task=task, obs=v.tolist(), action=float(np.asarray(a).reshape(-1)[0])))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "policy_golden.json")
json.dump(dict(generator="tools/gen_golden_policies.py", source="reference heuristic_policies.py executed here", cases=cases), open(out, "w"))
print("wrote", out, len(cases))
