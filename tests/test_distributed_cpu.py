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


# ---- collective="cabi": the control flow of the library's own row gather with a STUB library (no GPU here)
def _cabi_worker(rank, world, port, n_global, steps, q, depth, from_rank0=False):
    """The stub stands in for libjitterbug_hip.so's C ABI: step_rows_device / step_many_device write packed rows through raw pointers (the
    CPU oracle computes them), comm_unique_id / comm_init record the exchange, gather_rows_device / gather_block_device move `count` floats
    to rank 0 (over gloo, where the library uses grouped ncclSend / ncclRecv).  Streams and events are recorded, not executed: the test
    checks the ORDER of what ShardedJitterbugEnv issues (rotating buffers, the one-step-late gather, who waits for what)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from jitterbug_amd.distributed import ShardedJitterbugEnv
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = model.default_params()
    log = []

    def view(ptr, n):
        return np.ctypeslib.as_array((C.c_float * n).from_address(ptr))

    class StubLib:
        obs_dim = model.OBS_DIM["move_to_pose"]
        stream = 0

        def __init__(self, n, off):
            self.n, self.e = n, O.OracleEnv(n, "move_to_pose", P, seed=4, env_offset=off, step_limit=3, nsub=5)
            self.comm = None

        def reset(self):
            return self.e.reset()

        @staticmethod
        def comm_unique_id():
            return bytes(range(128))

        def comm_init(self, n_ranks, r, uid):
            assert uid == bytes(range(128)) and self.comm is None
            self.comm = (n_ranks, r)
            log.append(("comm_init", n_ranks, r))

        def comm_set_shards(self, sizes):
            assert self.comm is not None and list(sizes)[rank] == self.n          # (what jb_comm_set_shards refuses otherwise)
            self.nmax = max(sizes)
            log.append(("set_shards", tuple(int(x) for x in sizes)))

        def comm_destroy(self):
            assert self.comm is not None
            self.comm = None
            log.append(("comm_destroy",))

        def close(self):
            log.append(("close",))

        def scatter_actions_device(self, all_ptr, local_ptr, count, stream=None):
            assert self.comm == (world, rank) and count == self.nmax
            log.append(("scatter", count, stream))
            out = torch.empty(count, dtype=torch.float32)
            chunks = [torch.from_numpy(view(all_ptr, world * count).reshape(world, count)[r].copy()) for r in range(world)] if rank == 0 else None
            dist.scatter(out, chunks, src=0)
            view(local_ptr, count)[:] = out.numpy()

        def _rows(self, a):
            o, r, d = self.e.step(a)
            return np.concatenate([o, r[:, None], d[:, None].astype(np.float64)], 1).astype(np.float32)

        def step_rows_device(self, a_ptr, rows_ptr):
            view(rows_ptr, self.n * (self.obs_dim + 2))[:] = self._rows(view(a_ptr, self.n).copy()).ravel()
            log.append(("step", rows_ptr))

        def step_many_device(self, K, a_ptr, rows_ptr=None):
            W = self.obs_dim + 2
            tape = view(a_ptr, K * self.n).reshape(K, self.n).copy()
            out = view(rows_ptr, K * self.n * W).reshape(K, self.n, W)
            for k in range(K):
                out[k] = self._rows(tape[k])
            log.append(("step_many", K))

        def _gather(self, src_ptr, all_ptr, count, stream):
            assert self.comm == (world, rank)
            src = torch.from_numpy(view(src_ptr, count).copy())
            bufs = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
            dist.gather(src, bufs, dst=0)
            if rank == 0:
                view(all_ptr, world * count)[:] = torch.cat(bufs).numpy()

        def gather_rows_device(self, rows_ptr, all_ptr=None, stream=None):
            log.append(("gather", rows_ptr, stream))
            self._gather(rows_ptr, all_ptr, self.nmax * (self.obs_dim + 2), stream)          # blocks of the longest shard, like jb_gather_rows_device after jb_comm_set_shards

        def gather_block_device(self, src_ptr, all_ptr, count, stream=None):
            log.append(("gather_block", count, stream))
            self._gather(src_ptr, all_ptr, count, stream)

    class StubOps:                      # streams / events as labels: every call is logged
        def __init__(self):
            self.n_ev = 0

        def side_stream(self):
            return "side"

        def env_stream(self, env):
            return "env"

        def event(self):
            self.n_ev += 1
            return "ev%d" % self.n_ev

        def record(self, ev, stream):
            log.append(("record", ev, stream))

        def wait_event(self, stream, ev):
            log.append(("wait", stream, ev))

        def current_stream(self):
            return "cur"

        def raw(self, stream):
            return stream

        def synchronize(self, stream):
            log.append(("sync", stream))

    ops = StubOps()
    env = ShardedJitterbugEnv(n_global, "move_to_pose", seed=4, local_env_factory=lambda n, off: StubLib(n, off), pipeline_depth=depth, collective="cabi", stream_ops=ops)
    assert log[0] == ("comm_init", world, rank) and log[1] == ("set_shards", tuple(env.sizes))
    n_ev0 = ops.n_ev
    env.env.reset()
    rng = np.random.default_rng(0)
    outs = []
    for t in range(steps):
        ag = rng.uniform(-1, 1, size=n_global).astype(np.float32)
        if from_rank0:          # the whole round trip through the library: rank 0 brings every env's action, jb_scatter_actions_device hands them out
            r = env.step(ag if rank == 0 else None)
        else:
            r = env.step(local_actions=torch.from_numpy(ag)[env.lo:env.hi].contiguous())
        keep = lambda x: None if x is None else tuple(v.clone() for v in x)      # (results are views of the rotating receive buffers: valid until they come round again)
        if depth == 1:
            outs.append(keep(r))
        elif t > 0:
            outs.append(keep(r.get()))
    if depth == 2:
        outs.append(keep(env.flush().get()))
    tape = torch.from_numpy(rng.uniform(-1, 1, size=(4, n_global)).astype(np.float32))[:, env.lo:env.hi].contiguous()
    ro = env.rollout(4, local_actions=tape)
    assert ops.n_ev == n_ev0, "events are made once, with the buffers: %d were created while stepping" % (ops.n_ev - n_ev0)
    stub = env.env
    env.close()
    env.close()          # (twice is harmless)
    assert stub.comm is None and log[-2:] == [("comm_destroy",), ("close",)]
    if rank == 0:
        q.put(([(o.numpy(), r.numpy(), d.numpy()) for o, r, d in outs], (ro[0].numpy(), ro[1].numpy(), ro[2].numpy()), log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("depth,world,n_global,from_rank0", [(1, 2, 12, False), (2, 2, 12, False), (2, 2, 11, True), (1, 8, 4097, True), (2, 8, 41, True)])
def test_sharded_env_cabi_collective_control_flow_with_a_stub_library(depth, world, n_global, from_rank0):
    """ShardedJitterbugEnv(collective='cabi'): id exchange, jb_comm_init, jb_comm_set_shards, one jb_gather_rows_device per step (depth 2: one
    step late, on the side stream, behind the event of the step that wrote the rows; three rotating buffers, a buffer rewritten only after
    its gather's event), one jb_gather_block_device per rollout, no event made per step, close() - and the results equal one unsharded env.
    from_rank0: the actions travel through jb_scatter_actions_device (the whole round trip of a step through the library).  World 8 with
    uneven splits (4097 = 7 x 512 + 513; 41 = 1 x 6 + 7 x 5): ranks >= 2, blocks padded to the longest shard."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    steps = 5 if n_global < 1000 else 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cabi_worker, args=(r, world, port, n_global, steps, q, depth, from_rank0)) for r in range(world)]
    for p in procs:
        p.start()
    outs, ro, log = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = O.OracleEnv(n_global, "move_to_pose", model.default_params(), seed=4, step_limit=3, nsub=5)
    ref.reset()
    rng = np.random.default_rng(0)
    assert len(outs) == steps
    for (o, r, d) in outs:
        a = rng.uniform(-1, 1, size=n_global).astype(np.float32)
        eo, er, ed = ref.step(a)
        np.testing.assert_allclose(o, eo, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r, er, rtol=1e-5, atol=1e-6)
        assert np.array_equal(d, ed.astype(bool))
    tape = rng.uniform(-1, 1, size=(4, n_global)).astype(np.float32)
    for k in range(4):
        eo, er, ed = ref.step(tape[k])
        np.testing.assert_allclose(ro[0][k], eo, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ro[1][k], er, rtol=1e-5, atol=1e-6)
    # the order of what was issued (rank 0's log)
    i_steps = [i for i, e in enumerate(log) if e[0] == "step"]
    i_gathers = [i for i, e in enumerate(log) if e[0] == "gather"]
    assert len(i_steps) == steps and len(i_gathers) == steps and sum(e[0] == "gather_block" for e in log) == 1
    i_scatters = [i for i, e in enumerate(log) if e[0] == "scatter"]
    if from_rank0:          # one scatter per step, on the env's stream, right in front of the kernel that reads the block
        assert len(i_scatters) == steps and all(log[i][2] == "env" and log[i][1] == -(-n_global // world) for i in i_scatters)
        assert all(i_s < i_k and not any(log[j][0] == "step" for j in range(i_s, i_k)) for i_s, i_k in zip(i_scatters, i_steps))
    else:
        assert not i_scatters
    if depth == 2:
        bufs = [log[i][1] for i in i_steps]
        assert len(set(bufs[:3])) == 3 and bufs[3] == bufs[0] and bufs[4] == bufs[1]                 # three row buffers, rotating
        for k, i_g in enumerate(i_gathers):
            assert log[i_g][2] == "side" and log[i_g][1] == bufs[k]                                  # every gather on the side stream, of step k's buffer,
            if k + 1 < steps:
                assert i_g > i_steps[k + 1]                                                          # issued after step k + 1's kernel (one step late),
            assert log[i_g - 1][0] == "wait" and log[i_g - 1][1] == "side"                           # behind the event of the step that wrote the rows
            assert log[i_g + 1][0] == "record" and log[i_g + 1][2] == "side"                         # and an event behind it for its consumers
        assert log[i_steps[3] - 1][0] == "wait" and log[i_steps[3] - 1][1] == "env"                  # buffer 0 is rewritten only after gather 0's event
        assert log[i_steps[3] - 1][2] == log[i_gathers[0] + 1][1]
    else:
        assert all(log[i][2] == "env" for i in i_gathers)
