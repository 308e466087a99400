"""A wider run of the SURVEY 8(d) parity protocol than the test suite affords: teacher-forced (the oracle's state copied into the GPU env every
control step), N envs x 1000 control steps, every task, several seeds; ordinary and LEAN kernels; tipped-over regime.  Prints what
tests/test_gpu_parity.py asserts on, for the record (profiles/r04_parity_sweep.txt).   python tools/parity_sweep.py [n_envs] [n_seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _teacher_forced, MARGIN_TOL
from jitterbug_amd import model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
print("# teacher-forced parity, %d envs x 1000 control steps per (task, seed): entries outside 1e-4 rel + 1e-6 abs on WELL-conditioned env-steps (oracle contact-switch margin >= MARGIN_TOL = %.0f nm) [of those outside 1e-4 rel + 1e-5 abs: must be 0] / worst error there / share of ill-conditioned env-steps / share of ALL entries within tolerance / rewards within tolerance / Newton cap hits / excluded env-steps: worst error, cascade check (next step from the GPU's own state: checked, bad)" % (n, MARGIN_TOL * 1e9))
for flags, name in ((0, "ordinary kernel"), (2, "LEAN kernel")):
    for task in model.TASKS:
        for sd in range(seeds):
            r = _teacher_forced(task, n, 1000, seed=100 + 10 * model.TASKS.index(task) + sd, flags=flags)
            print("%-16s %-18s seed %3d : well-conditioned bad %d [strict %d] (worst %.1e, any > 1e-2: %d) | ill-conditioned env-steps %.5f | all entries %.6f | rewards %.6f | cap hits %.0f | excluded: worst %.1e, cascade %d checked %d bad | largest margin of a flipped env-step %.1f nm"
                  % (name, task, 100 + 10 * model.TASKS.index(task) + sd, r["well_bad"], r["strict_bad"], r["worst_well"], r["well_big"], r["ill_frac"], r["frac"], r["frac_reward"], r["cap"], r["worst_ill"], r["cascade_checked"], r["cascade_bad"], r["flip_margin_max"] * 1e9))
            sys.stdout.flush()
        if flags:
            break           # LEAN: one task is enough here (the suite checks it at 8192 envs too)
r = _teacher_forced("move_to_pose", n, 300, seed=55, flat_out=True, skip=250)
print("tipped regime (motor flat out, 250 lead-in steps, 300 compared): tipped %.2f | well-conditioned bad %d [strict %d] (worst %.1e) | ill %.5f | all entries %.6f | cap hits %.0f | excluded: worst %.1e, bad env-steps %d, cascade %d checked %d bad"
      % (r["tipped"], r["well_bad"], r["strict_bad"], r["worst_well"], r["ill_frac"], r["frac"], r["cap"], r["worst_ill"], r["ill_bad_steps"], r["cascade_checked"], r["cascade_bad"]))
print("# %.0f s" % (time.time() - t0))
