"""Worker of tests/test_gpu_surface.py::test_sharded_env_over_the_librarys_own_collective_one_rank: ShardedJitterbugEnv(collective="cabi") with a
world of ONE (the test box has one GPU; RCCL refuses two ranks on one device): jb_comm_init, the pipelined jb_gather_rows_device on a side
stream with three rotating buffers, the blocking form, and the fused rollout's jb_gather_block_device - real RCCL calls bound by the library,
no torch.distributed call on the data path (the gloo group only carries the communicator id) - against one plain env, bit for bit."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.distributed import ShardedJitterbugEnv
from jitterbug_amd.vec_env import JitterbugVecEnv

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
n, task = 1000, "move_to_pose"
for depth in (1, 2):
    sh = ShardedJitterbugEnv(n, task, seed=4, device="cuda:0", pipeline_depth=depth, collective="cabi")
    sh.env.reset_device()
    whole = JitterbugVecEnv(n, task, seed=4)
    whole.reset()
    rng = np.random.default_rng(0)
    expected = []
    for t in range(9):
        acts = rng.uniform(-1, 1, size=n).astype(np.float32)
        # odd steps: the whole round trip through the library - rank 0 brings every env's action (host array, or a device tensor made on the
        # caller's stream) and jb_scatter_actions_device hands them out; even steps: actions already resident on the rank
        res = sh.step(acts if t % 4 == 1 else torch.as_tensor(acts, device="cuda:0")) if t % 2 else sh.step(local_actions=torch.as_tensor(acts, device="cuda:0"))
        expected.append(whole.step(acts))
        if depth == 2:
            if t == 0:
                assert res is None
                continue
            res, want = res.get(), expected[t - 1]
        else:
            want = expected[t]
        o1, r1, d1 = (x.cpu().numpy() for x in res)
        assert np.array_equal(o1, want[0]) and np.array_equal(r1, want[1]) and np.array_equal(d1, want[2].astype(bool)), "depth %d step %d differs" % (depth, t)
    if depth == 2:
        o1, r1, d1 = (x.cpu().numpy() for x in sh.flush().get())
        assert np.array_equal(o1, expected[-1][0]) and np.array_equal(r1, expected[-1][1])
    # fused rollout: one launch, ONE block gather
    K = 7
    tape = rng.uniform(-1, 1, size=(K, n)).astype(np.float32)
    o, r, d = (x.cpu().numpy() for x in sh.rollout(K, local_actions=torch.as_tensor(tape, device="cuda:0")))
    for k in range(K):
        w = whole.step(tape[k])
        assert np.array_equal(o[k], w[0]) and np.array_equal(r[k], w[1]) and np.array_equal(d[k], w[2].astype(bool)), "rollout step %d differs" % k
    sh.close(); sh.close(); whole.close()          # close(): flush, both streams idle, jb_comm_destroy, then the env (twice is harmless)
    assert sh.env is None
try:
    ShardedJitterbugEnv(n, task, seed=4, device="cuda:0", flags=2)          # variant="auto" + JB_FLAG_LEAN in flags: refused, not silently cleared
    raise SystemExit("variant='auto' with JB_FLAG_LEAN in flags was accepted")
except ValueError as e:
    assert "JB_FLAG_LEAN" in str(e)
print("CABI_SHARDED_OK", flush=True)
dist.destroy_process_group()
