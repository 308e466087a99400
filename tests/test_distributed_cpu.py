"""World-size-2 gloo tests of the sharding layer (runs on CPU).  The local stepper is the CPU oracle standing in for
the GPU env: what is under test is the partition, the action scatter, the row gather and the invariance of results to
the split (global env indices key the RNG streams)."""
import os
import socket

import numpy as np
import pytest

from jitterbug_amd import model
from jitterbug_amd.distributed import shard_range


def test_shard_range_partition():
    for n in (1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, steps, q, depth=1):
    import torch
    import torch.distributed as dist
    from jitterbug_amd.distributed import ShardedJitterbugEnv
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = model.default_params()

    class Local:
        def __init__(self, n, off):
            self.e = O.OracleEnv(n, "move_to_pose", P, seed=4, env_offset=off, step_limit=3, nsub=5)

        def reset(self):
            return self.e.reset()

        def step(self, a):
            return self.e.step(a)

    env = ShardedJitterbugEnv(n_global, "move_to_pose", seed=4, local_env_factory=lambda n, off: Local(n, off), pipeline_depth=depth)
    outs = [env.reset()]
    rng = np.random.default_rng(0)
    for t in range(steps):
        acts = rng.uniform(-1, 1, size=n_global) if rank == 0 else None
        r = env.step(acts)
        if depth == 1:
            outs.append(r)
        elif t > 0:                             # depth 2: a call hands back the previous step's results (a PendingRows handle)
            outs.append(r.get())
        else:
            assert r is None
    if depth == 2:
        outs.append(env.flush().get())
        assert env.flush() is None
    if rank == 0:
        q.put((outs[0].numpy(), [(o.numpy(), r.numpy(), d.numpy()) for o, r, d in outs[1:]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("depth", [1, 2])
def test_sharded_env_matches_single_process_gloo(depth):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    n_global, steps, world = 11, 5, 2                     # uneven split: 6 + 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, steps, q, depth)) for r in range(world)]
    for p in procs:
        p.start()
    obs0, outs = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = O.OracleEnv(n_global, "move_to_pose", model.default_params(), seed=4, step_limit=3, nsub=5)
    np.testing.assert_allclose(obs0, ref.reset(), rtol=1e-6, atol=1e-7)
    rng = np.random.default_rng(0)
    saw_done = False
    for (o, r, d) in outs:
        a = rng.uniform(-1, 1, size=n_global).astype(np.float32)     # actions travel as fp32
        ro, rr, rd = ref.step(a)
        np.testing.assert_allclose(o, ro, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r, rr, rtol=1e-5, atol=1e-6)
        assert np.array_equal(d, rd.astype(bool))
        saw_done |= bool(d.any())
    assert saw_done            # the auto-reset (episode 2 streams) crossed the shard boundary consistently
