"""Episode returns of the reference's heuristic policies (heuristic_policies.py, restated) on the GPU simulator, per task.
External behavioural check of the physics: the reference's paper says these policies solve the tasks and calls a task
solved at a return of about 900 (manuscript/ICRA2020/root.tex:279, 392)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.evaluate_policy import evaluate_heuristic_batch
from jitterbug_amd import model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for task in model.TASKS:
    r = np.asarray(evaluate_heuristic_batch(task, num_repeats=n, seed=0)).sum(axis=1)
    print("%-18s episode return: mean %.1f median %.1f p10 %.1f p90 %.1f  frac>=900 %.3f  frac>=800 %.3f" % (task, r.mean(), np.median(r), np.quantile(r, .1), np.quantile(r, .9), (r >= 900).mean(), (r >= 800).mean()), flush=True)
