"""Policy evaluation (reference: benchmarks/evaluate_policy.py:10-35).

``evaluate_policy(env, policy, num_repeats=...)`` keeps the reference's signature and result: a
``num_repeats x (env._step_limit - 1)`` array of per-step rewards from running ``policy(ts)`` on the single env.
``evaluate_heuristic_batch`` is the MI355X way to get the same statistic: one repeat per env of a batch, the reference's
heuristic policy evaluated on the device, and all ``_step_limit - 1`` steps chained on the GPU without host round trips.

Callers that also use torch in the same process may create handles and torch tensors in either order: jitterbug_amd._lib binds to the
HIP runtime PyTorch-ROCm ships before it loads the library, so the process holds one runtime whatever the import order
(tests/test_gpu_surface.py::test_handles_and_torch_tensors_in_either_order)."""
import numpy as np


def evaluate_policy(env, policy, *, num_repeats=20, verbose=False):
    if verbose:
        print("Evaluating {} on {}".format(policy, env.task.task))
    n = int(env._step_limit - 1)
    results = np.empty((num_repeats, n))
    for repeat in range(num_repeats):
        ts = env.reset()
        for i in range(n):
            ts = env.step(policy(ts))
            results[repeat, i] = ts.reward
    return results


def evaluate_heuristic_batch(task, num_repeats=20, seed=0, time_limit=10, device_id=0):
    """-> rewards [num_repeats, step_limit - 1] with the task's heuristic policy (one repeat per env of a batch)."""
    from .vec_env import JitterbugVecEnv
    env = JitterbugVecEnv(num_repeats, task, seed=seed, device_id=device_id, time_limit=time_limit, auto_reset=False)
    env.reset()
    rew, _ = env.rollout_policy(env.step_limit - 1)
    env.close()
    return rew.T.astype(np.float64)
