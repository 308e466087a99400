"""Generates tests/golden/norm_tables.json: the literal tables and constants of the reference task that define the observation
normalisation and the episode structure, extracted from the reference's source by the AST (the module itself cannot be
imported here: dm_control / MuJoCo are not installed - an ordinary ModuleNotFoundError).

  Jitterbug._NORM_ALL, Jitterbug._NORM_TASKS        reference jitterbug_dmc/jitterbug.py:324-372
  DEFAULT_TIME_LIMIT, DEFAULT_CONTROL_TIMESTEP, TARGET_SPEED   reference jitterbug.py:56-58

Only the VALUES of those assignments are evaluated (numpy in scope for np.array / np.pi); no other reference code runs.
    python tools/gen_golden_norm.py
"""
import ast
import json
import os

import numpy as np

SRC = "/root/reference/jitterbug_dmc/jitterbug.py"
tree = ast.parse(open(SRC).read())


def value_of(node):
    return eval(compile(ast.Expression(node.value), SRC, "eval"), {"np": np, "dict": dict})


out = {"generator": "tools/gen_golden_norm.py", "source": "reference jitterbug_dmc/jitterbug.py (AST-extracted literals)"}
for node in tree.body:
    if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in ("DEFAULT_TIME_LIMIT", "DEFAULT_CONTROL_TIMESTEP", "TARGET_SPEED"):
        out[node.targets[0].id] = float(value_of(node))
    if isinstance(node, ast.ClassDef) and node.name == "Jitterbug":
        for sub in node.body:
            if isinstance(sub, ast.Assign) and isinstance(sub.targets[0], ast.Name) and sub.targets[0].id == "_NORM_ALL":
                out["_NORM_ALL"] = np.asarray(value_of(sub), dtype=np.float64).tolist()
            if isinstance(sub, ast.Assign) and isinstance(sub.targets[0], ast.Name) and sub.targets[0].id == "_NORM_TASKS":
                out["_NORM_TASKS"] = {k: np.asarray(v, dtype=np.float64).reshape(-1, 2).tolist() for k, v in value_of(sub).items()}
assert set(out) >= {"_NORM_ALL", "_NORM_TASKS", "DEFAULT_TIME_LIMIT", "DEFAULT_CONTROL_TIMESTEP", "TARGET_SPEED"}, sorted(out)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "norm_tables.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, {k: (np.shape(v) if not isinstance(v, (dict, float, str)) else v if not isinstance(v, dict) else {a: np.shape(b) for a, b in v.items()}) for k, v in out.items()})
