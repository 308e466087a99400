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


def _variant_worker(rank, world, port, q):
    import torch.distributed as dist
    from jitterbug_amd.distributed import ShardedJitterbugEnv
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Stub:
        obs_dim = 15

        def __init__(self, n, off):
            self.n, self.off = n, off

    out = []
    for n_global, kw in ((8193, {}), (16383, dict(per_env_model=True)), (8192, {}), (16382, dict(per_env_model=True)), (8193, dict(variant="ordinary"))):
        env = ShardedJitterbugEnv(n_global, "move_from_origin", local_env_factory=lambda n, off: Stub(n, off), **kw)
        out.append((n_global, tuple(sorted(kw.items())), env.n_local, env.variant))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_every_rank_resolves_the_same_kernel_variant_gloo():
    """variant='auto' is resolved from the GLOBAL batch and the world size, never from a rank's own shard length: with 8193 envs on two
    ranks (4097 + 4096) a per-shard rule would run the LEAN kernel on rank 0 and the ordinary one on rank 1 - two roundings in one
    batch.  (The reference's harness builds its vec env with no knobs: benchmarks/benchmark.py:146-171.)"""
    import torch.multiprocessing as mp
    from jitterbug_amd import variants
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_variant_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1 = got[0], got[1]
    assert [x[3] for x in r0] == [x[3] for x in r1] == ["lean", "lean", "ordinary", "ordinary", "ordinary"]
    assert r0[0][2] == 4097 and r1[0][2] == 4096 and r0[1][2] == 8192 and r1[1][2] == 8191          # ... although the shards straddle the thresholds
    # the thresholds themselves, in the one place they live
    assert variants.resolve("auto", 4096) == "ordinary" and variants.resolve("auto", 4097) == "lean"
    assert variants.resolve("auto", 8191, per_env_model=True) == "ordinary" and variants.resolve("auto", 8192, per_env_model=True) == "lean"
    assert variants.flags_for("lean", 64, flags=8) == 10 and variants.flags_for("ordinary", 10 ** 6, flags=2 | 16) == 16
    with pytest.raises(ValueError):
        variants.resolve("fast", 64)


def _rollout_worker(rank, world, port, n_global, K, q):
    import torch
    import torch.distributed as dist
    from jitterbug_amd.distributed import ShardedJitterbugEnv
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = model.default_params()

    class Local:
        obs_dim = 19

        def __init__(self, n, off):
            self.e = O.OracleEnv(n, "move_to_pose", P, seed=4, env_offset=off, step_limit=3, nsub=5)

        def reset(self):
            return self.e.reset()

        def step(self, a):
            return self.e.step(a)

    env = ShardedJitterbugEnv(n_global, "move_to_pose", seed=4, local_env_factory=lambda n, off: Local(n, off))
    obs0 = env.reset()
    rng = np.random.default_rng(0)
    tape = rng.uniform(-1, 1, size=(K, n_global)).astype(np.float32) if rank == 0 else None
    out = env.rollout(K, actions_global=tape)
    if rank == 0:
        q.put((obs0.numpy(), [t.numpy() for t in out]))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_rollout_one_gather_of_k_steps_matches_single_process_gloo():
    """ShardedJitterbugEnv.rollout(K): K steps on every shard, then ONE gather of the [K, N_local, D+2] blocks (ragged shards: 6 + 5) -
    equal to K steps of one unsharded env, episode ends inside the rollout included."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    n_global, K, world = 11, 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rollout_worker, args=(r, world, port, n_global, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    obs0, (ob, rw, dn) = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ob.shape == (K, n_global, 19) and rw.shape == (K, n_global) and dn.shape == (K, n_global)
    ref = O.OracleEnv(n_global, "move_to_pose", model.default_params(), seed=4, step_limit=3, nsub=5)
    np.testing.assert_allclose(obs0, ref.reset(), rtol=1e-6, atol=1e-7)
    rng = np.random.default_rng(0)
    tape = rng.uniform(-1, 1, size=(K, n_global)).astype(np.float32)
    for k in range(K):
        ro, rr, rd = ref.step(tape[k])
        np.testing.assert_allclose(ob[k], ro, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rw[k], rr, rtol=1e-5, atol=1e-6)
        assert np.array_equal(dn[k], rd.astype(bool))
    assert dn.sum() == 2 * n_global
