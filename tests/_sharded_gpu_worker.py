"""Worker of tests/test_gpu_surface.py::test_sharded_env_two_ranks_one_gpu (launched by torch.distributed.run, 2 or 4 ranks on cuda:0, gloo)."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.distributed import ShardedJitterbugEnv
from jitterbug_amd.vec_env import JitterbugVecEnv

dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
n, task = 1003, "move_to_pose"
sh = ShardedJitterbugEnv(n, task, seed=4, device="cuda:0")
ob = sh.reset()
rng = np.random.default_rng(0)
whole = None
if rank == 0:
    whole = JitterbugVecEnv(n, task, seed=4)
    ow = whole.reset()
    assert np.array_equal(ob.cpu().numpy(), ow), "reset differs"
for t in range(6):
    acts = rng.uniform(-1, 1, size=n).astype(np.float32) if rank == 0 else None
    res = sh.step(acts)
    if rank == 0:
        o2, r2, d2, _ = whole.step(acts)
        o1, r1, d1 = (x.cpu().numpy() for x in res)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2.astype(bool)), "step %d differs" % t
    else:
        assert res is None
sh.close(); sh.close()
assert sh.env is None
# the pipelined product path (what bench.py --gpus N times): results arrive one call late, bit-identical
sh2 = ShardedJitterbugEnv(n, task, seed=4, device="cuda:0", pipeline_depth=2)
sh2.reset()
whole2 = None
if rank == 0:
    whole2 = JitterbugVecEnv(n, task, seed=4)
    whole2.reset()
rng = np.random.default_rng(0)
expected = []
lo, hi = sh2.lo, sh2.hi
for t in range(7):
    acts = rng.uniform(-1, 1, size=n).astype(np.float32)                    # every rank draws the same stream and passes ITS slice
    res = sh2.step(local_actions=torch.as_tensor(acts[lo:hi], device="cuda:0"))
    if rank == 0:
        expected.append(whole2.step(acts))
        if t == 0:
            assert res is None
        else:
            o2, r2, d2, _ = expected[t - 1]
            o1, r1, d1 = (x.cpu().numpy() for x in res.get())
            assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2.astype(bool)), "pipelined step %d differs" % t
    elif res is not None:
        assert res.get() is None
stale = sh2.step(local_actions=torch.zeros(hi - lo, device="cuda:0"))     # a handle nobody asks for in time
if rank == 0:
    expected.append(whole2.step(np.zeros(n, dtype=np.float32)))
for t in range(4):
    sh2.step(local_actions=torch.zeros(hi - lo, device="cuda:0"))
    if rank == 0:
        expected.append(whole2.step(np.zeros(n, dtype=np.float32)))
try:
    stale.get()
    raise SystemExit("an expired PendingRows handle did not raise")
except RuntimeError as e:
    assert "expired" in str(e)
res = sh2.flush()
if rank == 0:
    o2, r2, d2, _ = expected[-1]
    o1, r1, d1 = (x.cpu().numpy() for x in res.get())
    assert np.array_equal(o1, o2) and np.array_equal(r1, r2), "pipelined flush differs"
sh2.close()
# the fused rollout across shards: K steps in one launch per rank, ONE gather of [K, N_local, D+2]; tape from rank 0, per-rank tapes,
# and the in-kernel heuristic policy - each bit-identical to the same K steps of one unsharded env
K = 9
for mode in ("global_tape", "local_tape", "policy"):
    sh3 = ShardedJitterbugEnv(n, task, seed=4, device="cuda:0", time_limit=0.05)       # 5-step episodes: an auto-reset inside the rollout
    sh3.reset()
    rng = np.random.default_rng(3)
    tape = rng.uniform(-1, 1, size=(K, n)).astype(np.float32)
    if mode == "global_tape":
        res = sh3.rollout(K, actions_global=tape if rank == 0 else None)
    elif mode == "local_tape":
        res = sh3.rollout(K, local_actions=torch.as_tensor(tape[:, sh3.lo:sh3.hi], device="cuda:0").contiguous())
    else:
        res = sh3.rollout(K)
    if rank == 0:
        w = JitterbugVecEnv(n, task, seed=4, time_limit=0.05)
        ob_w = w.reset()
        o1, r1, d1 = (x.cpu().numpy() for x in res)
        assert o1.shape == (K, n, w.obs_dim) and d1.sum() == n
        for k in range(K):
            a = tape[k] if mode != "policy" else w.policy(ob_w)
            ob_w, r2, d2, _ = w.step(a)
            assert np.array_equal(o1[k], ob_w) and np.array_equal(r1[k], r2) and np.array_equal(d1[k], d2.astype(bool)), "rollout (%s) step %d differs" % (mode, k)
        w.close()
    else:
        assert res is None
    sh3.close()
dist.barrier()
if rank == 0:
    print("SHARDED_OK", flush=True)
dist.destroy_process_group()
