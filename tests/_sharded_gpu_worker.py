"""Worker of tests/test_gpu_surface.py::test_sharded_env_two_ranks_one_gpu (launched by torch.distributed.run, 2 ranks on cuda:0, gloo)."""
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
dist.barrier()
if rank == 0:
    print("SHARDED_OK", flush=True)
dist.destroy_process_group()
