"""GPU tests of the reference-shaped Python surface: suite.load / Environment / Physics accessors / JitterbugGymEnv."""
import collections

import numpy as np
import pytest

from jitterbug_amd import model

pytestmark = pytest.mark.gpu


def test_suite_load_reset_step_specs():
    from jitterbug_amd import suite
    from jitterbug_amd.specs import StepType
    from oracle import oracle as O
    P = model.default_params()
    for task, D in model.OBS_DIM.items():
        env = suite.load("jitterbug", task, task_kwargs=dict(random=3, norm_obs=True))
        spec = env.action_spec()
        assert spec.shape == (1,) and spec.minimum[0] == -1 and spec.maximum[0] == 1 and spec.dtype == np.float64
        ospec = env.observation_spec()
        assert isinstance(ospec, collections.OrderedDict) and sum(s.shape[0] for s in ospec.values()) == D
        assert list(ospec)[:4] == ["position", "velocity", "motor_position", "motor_velocity"]
        ts = env.reset()
        assert ts.step_type == StepType.FIRST and ts.reward is None and ts.discount is None
        assert ts.observation["motor_position"].shape == (1,) and ts.observation["position"].shape == (7,)
        ts = env.step(0.8)
        assert ts.step_type == StepType.MID and 0.0 <= ts.reward <= 1.0 and ts.discount == 1.0
        # Physics accessors against the oracle's observation/reward on the same state
        ph = env.physics
        q, v = ph.qpos(), ph.qvel()
        tgt = np.array([ph.target_position_xyz()[0], ph.target_position_xyz()[1], ph.target_direction_yaw() % (2 * np.pi)])
        flat = np.concatenate(list(ts.observation.values()))
        np.testing.assert_allclose(flat, O.observation(P, task, q, v, tgt), rtol=1e-5, atol=2e-6)
        assert abs(ts.reward - O.reward(P, task, q, v, tgt)) < 1e-5
        assert ph.jitterbug_position_xyz().shape == (3,) and ph.jitterbug_position_quat().shape == (4,)
        assert ph.angle_jitterbug_to_target().shape == (1,) and -np.pi < ph.angle_jitterbug_to_target()[0] <= np.pi
        assert -np.pi < ph.motor_position() <= np.pi
        np.testing.assert_allclose(np.linalg.norm(ph.target_position_in_jitterbug_frame()),
                                   np.linalg.norm(ph.target_position_xyz() - ph.jitterbug_position_xyz()), rtol=1e-5)   # fp32 quaternion: unit to ~1e-7
        env.close()


def test_episode_end_semantics_and_flat_observation():
    from jitterbug_amd import suite
    env = suite.load("jitterbug", "move_from_origin", task_kwargs=dict(random=0, time_limit=0.05),
                     environment_kwargs=dict(flat_observation=True))
    assert env._step_limit == pytest.approx(5.0) and env.control_timestep() == 0.01
    ts = env.reset()
    assert list(ts.observation) == ["observations"] and ts.observation["observations"].shape == (15,)
    for t in range(4):
        assert env.step([0.5]).mid()
    ts = env.step(np.array([0.5]))
    assert ts.last() and ts.discount == 1.0
    ts = env.step(0.5)                       # a step after LAST is a reset
    assert ts.first() and ts.reward is None
    env.close()


def test_gym_wrapper():
    from jitterbug_amd import JitterbugGymEnv, suite
    env = JitterbugGymEnv(suite.load("jitterbug", "move_to_pose", task_kwargs=dict(random=1),
                                     environment_kwargs=dict(flat_observation=True)))
    assert env.num_envs == 1
    assert env.action_space.shape == (1,) and env.action_space.low[0] == -1 and env.action_space.dtype == np.float32
    assert env.observation_space["observations"].shape == (19,)
    obs = env.reset()
    assert obs["observations"].shape == (19,)
    total = 0.0
    for t in range(20):
        obs, r, done, info = env.step(env.action_space.sample())
        total += r
        assert not done and info == {}
        env.render()
    assert env.frame_count == 20 and np.isfinite(total)
    env.close()


def test_vec_env_shapes_and_done_flags():
    from jitterbug_amd import JitterbugVecEnv
    env = JitterbugVecEnv(100, "face_direction", seed=2, time_limit=0.03)
    obs = env.reset()
    assert obs.shape == (100, 16) and obs.dtype == np.float32
    env.step_async(np.zeros(100))
    obs, rew, done, infos = env.step_wait()
    assert rew.shape == (100,) and done.dtype == bool and not done.any()
    env.step(np.zeros(100))
    _, _, done, _ = env.step(np.zeros(100))
    assert done.all()
    env.close()


def test_step_async_overlaps_host_work_and_equals_step():
    """step_async / step_wait are two real halves (reference benchmarks/benchmark.py:146-171: SubprocVecEnv's workers step while the
    trainer goes on): 200 steps with 2 ms of host work between the halves must take max(host, device) - not their sum - within 5 %,
    and hand out the very arrays step() returns.  Views of the pinned buffers (copy=False) and MonitoredVecEnv ride the same path."""
    import time
    from jitterbug_amd.vec_env import JitterbugVecEnv, MonitoredVecEnv
    n, steps, host_s = 4096, 200, 0.002
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(steps, n)).astype(np.float32)
    a_env, s_env = JitterbugVecEnv(n, "move_from_origin", seed=3), JitterbugVecEnv(n, "move_from_origin", seed=3)
    a_env.reset(), s_env.reset()

    def busy(seconds):
        t = time.perf_counter()
        while time.perf_counter() - t < seconds:
            pass

    for k in range(20):          # warm-up (first launch, pinned allocation); both envs stay in step
        a_env.step_async(acts[k]); ra = a_env.step_wait(); rs = s_env.step(acts[k])
        assert all(np.array_equal(x, y) for x, y in zip(ra[:3], rs[:3]))
    with pytest.raises(RuntimeError):
        a_env.step_wait()                                        # no step pending
    a_env.step_async(acts[0])
    with pytest.raises(Exception):
        a_env.step(acts[0])                                      # jb_step refuses while a step is in flight
    a_env.step_wait(); s_env.step(acts[0])
    dev_t = []
    for k in range(20, steps):
        t0 = time.perf_counter()
        s_env.step(acts[k])
        dev_t.append(time.perf_counter() - t0)
    dev_s = float(np.median(dev_t))                              # one synchronous step, device + copies (medians: a shared box preempts now and then)
    outs, both_t = [], []
    for k in range(20, steps):
        t0 = time.perf_counter()
        a_env.step_async(acts[k])
        busy(host_s)                                             # the caller's own work (a policy's forward pass, say)
        o, r, d, _ = a_env.step_wait(copy=False)
        both_t.append(time.perf_counter() - t0)
        outs.append((o.copy(), r.copy(), d.copy()))
    both_s = float(np.median(both_t))
    assert both_s <= 1.05 * max(host_s, dev_s) + 2e-5, "async step %.3f ms vs host %.3f ms / device %.3f ms: the halves add up instead of overlapping" % (1e3 * both_s, 1e3 * host_s, 1e3 * dev_s)
    assert both_s < 0.8 * (host_s + dev_s), (both_s, host_s, dev_s)
    ref = JitterbugVecEnv(n, "move_from_origin", seed=3)
    ref.reset()
    for k in list(range(20)) + [0]:
        ref.step(acts[k])
    for k in range(20, steps):
        o, r, d, _ = ref.step(acts[k])
        assert np.array_equal(o, outs[k - 20][0]) and np.array_equal(r, outs[k - 20][1]) and np.array_equal(d, outs[k - 20][2]), k
    m = MonitoredVecEnv(JitterbugVecEnv(16, "move_from_origin", seed=1, time_limit=0.04))
    m.reset()
    for t in range(4):
        m.step_async(np.full(16, 0.5))
        obs, rew, done, infos = m.step_wait()
    assert done.all() and all(i["episode"]["l"] == 4 for i in infos)
    for e in (a_env, s_env, ref, m):
        e.close()


def test_monitored_vec_env(tmp_path):
    from jitterbug_amd.vec_env import JitterbugVecEnv, MonitoredVecEnv
    env = MonitoredVecEnv(JitterbugVecEnv(10, "move_from_origin", seed=1, time_limit=0.04), filename=str(tmp_path / "run"))
    env.reset()
    seen = 0
    for t in range(9):
        obs, rew, done, infos = env.step(np.full(10, 0.5))
        if done.any():
            assert done.all() and all(i["episode"]["l"] == 4 for i in infos)
            seen += 1
        else:
            assert all(i == {} for i in infos)
    assert seen == 2 and len(env.episode_returns) == 20
    env.close()
    lines = open(str(tmp_path / "run.monitor.csv")).read().splitlines()
    assert lines[0].startswith("#{") and lines[1] == "r,l,t" and len(lines) == 22


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_env_two_ranks_one_gpu(world):
    """The N>1 data path (scatter -> jb_step_rows_device -> gather) with two - and four: ranks >= 2, an uneven 251 / 251 / 251 / 250 split -
    ranks sharing cuda:0 over gloo: bit-identical to one unsharded env, both pipeline depths, fused rollouts, close().  (RCCL needs one GPU
    per rank, which the test box does not have; the collectives are staged.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29533 + world), os.path.join(root, "tests", "_sharded_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("torch_first", [True, False])
def test_handles_and_torch_tensors_in_either_order(torch_first):
    """INTEGRATION.md 3 / evaluate_policy.py: a process holds ONE HIP runtime whatever the order in which torch touches the GPU and a handle
    is created (jitterbug_amd._lib preloads the copy PyTorch-ROCm ships).  Each order in a fresh process: a step through device pointers of
    torch tensors, and a torch kernel on the result."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        torch_first = %r
        if torch_first:
            import torch
            x = torch.ones(8, device="cuda:0") * 2
            assert float(x.sum()) == 16.0
        from jitterbug_amd.vec_env import JitterbugVecEnv
        env = JitterbugVecEnv(64, "move_from_origin", seed=2)
        env.reset()
        import torch
        dev = torch.device("cuda", 0)
        a = torch.zeros(64, device=dev); obs = torch.zeros((64, env.obs_dim), device=dev); rew = torch.zeros(64, device=dev); done = torch.zeros(64, device=dev, dtype=torch.uint8)
        torch.cuda.synchronize()
        env.step_device(a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr()); env.synchronize()
        assert bool(torch.isfinite(obs).all()) and float(obs.abs().sum()) > 0
        env.close()
        print("ORDER_OK")
    """) % (root, torch_first)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ORDER_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_sharded_env_over_the_librarys_own_collective_one_rank():
    """ShardedJitterbugEnv(collective='cabi') - the product class on the C ABI's own RCCL gather - with a world of one on the
    one-GPU box: both pipeline depths and the fused rollout, bit-identical to one plain env (tests/_sharded_cabi_worker.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_sharded_cabi_worker.py")], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CABI_SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_integration_md_stub_runs_and_matches_vec_env():
    """The ctypes stub printed in INTEGRATION.md is executed as written (only the library path is substituted)."""
    import os
    import numpy as np
    from jitterbug_amd import _lib
    from jitterbug_amd.vec_env import JitterbugVecEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "INTEGRATION.md")).read()
    a = src.index("```python") + len("```python")
    code = src[a:src.index("```", a)].replace('C.CDLL("libjitterbug_hip.so")', 'C.CDLL(%r)' % _lib.LIB_PATH)
    _lib.load()                                   # same HIP runtime preload as the package
    ns = {}
    exec(code, ns)
    b = ns["HipBatch"](33, "move_to_position", seed=9)
    v = JitterbugVecEnv(33, "move_to_position", seed=9)
    assert np.array_equal(b.reset(), v.reset())
    rng = np.random.default_rng(0)
    for _ in range(3):
        act = rng.uniform(-1, 1, size=33).astype(np.float32)
        o1, r1, d1, _ = b.step(act)
        o2, r2, d2, _ = v.step(act)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2.astype(bool))
    act = rng.uniform(-1, 1, size=33).astype(np.float32)
    b.step_async(act)                             # the stub's step_async / step_wait pair = one step
    o1, r1, d1, _ = b.step_wait()
    o2, r2, d2, _ = v.step(act)
    assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2.astype(bool))
    # the second block of the document: the fused rollout stub (jb_step_many_device), as written
    a2 = src.index("```python", a) + len("```python")
    exec(src[a2:src.index("```", a2)], ns)
    import torch
    K = 12
    rew = torch.zeros((K, 33), device="cuda:0"); obs = torch.zeros((33, b.D), device="cuda:0"); done = torch.zeros((33,), device="cuda:0", dtype=torch.uint8)
    ns["rollout"](b, K, None, rew.data_ptr(), obs.data_ptr(), done.data_ptr())          # the heuristic policy acts inside the kernel
    r2, o2 = v.rollout_policy(K)
    assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(obs.cpu().numpy(), o2)
    v.close()


def test_task_object_methods_of_the_reference():
    """Jitterbug.get_observation / get_reward / the four reward terms / initialize_episode take a `physics`, like the reference
    (jitterbug.py:601, 673, 840-925); here they run the GPU kernels (jb_observe, jb_reward_terms, jb_reset) and must agree with the
    oracle evaluated on the state the Physics accessors report."""
    from jitterbug_amd import suite
    from oracle import oracle as O
    P = model.default_params()
    for task in model.TASKS:
        env = suite.load("jitterbug", task, task_kwargs=dict(random=5))
        env.reset()
        for _ in range(3):
            ts = env.step(0.7)
        ph, tk = env.physics, env.task
        calls0 = _count_state_fetches(ph)
        q, v = ph.qpos(), ph.qvel()
        tgt = np.array([ph.target_position_xyz()[0], ph.target_position_xyz()[1], ph.target_direction_yaw() % (2 * np.pi)])
        obs = tk.get_observation(ph)
        assert isinstance(obs, collections.OrderedDict) and list(obs) == list(ts.observation)
        np.testing.assert_allclose(np.concatenate(list(obs.values())), O.observation(P, task, q, v, tgt), rtol=1e-5, atol=2e-6)
        for k in obs:
            np.testing.assert_allclose(obs[k], ts.observation[k], rtol=1e-6, atol=1e-8)   # the observe kernel and the step kernel are compiled separately: same formulas, multiply-adds may fuse differently
        assert tk.get_reward(ph) == pytest.approx(ts.reward, abs=1e-7)
        terms = O.reward_terms(P, q, v, tgt)
        assert tk.position_reward(ph) == pytest.approx(terms["P"], abs=1e-5)
        assert tk.heading_reward(ph) == pytest.approx(terms["H"], abs=1e-5)
        assert tk.velocity_reward(ph) == pytest.approx(terms["V"], abs=1e-4)
        assert tk.upright_reward(ph) == pytest.approx(terms["U"], abs=1e-5)
        # the reference's sensor through physics.named, and the named state views
        np.testing.assert_allclose(ph.named.data.sensordata["jitterbug_framelinvel"], ph.jitterbug_framelinvel())
        np.testing.assert_array_equal(ph.named.data.qpos["root"], q[:7])
        assert ph.named.data.qvel["jointMass"][0] == v[14] and ph.named.data.xmat["jitterbug", "zz"] == pytest.approx(ph.upright())
        np.testing.assert_allclose(ph.named.data.geom_xpos["target"], ph.target_position_xyz())
        # all of the above cost ONE state fetch (cached per simulator change)
        assert _count_state_fetches(ph) - calls0 <= 1
        # initialize_episode: a new episode for this env (velocities zero, new target for the target tasks)
        tk.initialize_episode(ph)
        assert np.all(ph.qvel() == 0) and ph.jitterbug_position_xyz()[2] == pytest.approx(0.035, abs=1e-6)
        if task in ("move_to_position", "move_to_pose"):
            assert 0.05 - 1e-6 <= np.hypot(*ph.target_position_xyz()[:2]) < 0.2 + 1e-6
            assert not np.allclose(ph.target_position_xyz()[:2], tgt[:2])
        env.close()


def _count_state_fetches(ph):
    """number of device state fetches the env has served (instrumented once per Physics object)"""
    venv = ph._venv
    if not hasattr(venv, "_fetches"):
        venv._fetches = 0
        orig = venv.get_state

        def counted():
            venv._fetches += 1
            return orig()
        venv.get_state = counted
    return venv._fetches


def test_vec_env_infos_is_always_a_list():
    from jitterbug_amd.vec_env import JitterbugVecEnv
    for n in (3, 64, 65, 300):
        env = JitterbugVecEnv(n, "move_from_origin")
        env.reset()
        _, _, _, infos = env.step(np.zeros(n, dtype=np.float32))
        assert isinstance(infos, list) and len(infos) == n and infos[n - 1] == {}
        env.close()


def test_velocity_in_target_frame_uses_the_body_com_sensor():
    """move_in_direction with a spinning robot: obs[16:19] and the velocity reward follow framelinvel at the root body's own COM
    (jitterbug.xml:121, objtype="body"), GPU vs oracle."""
    from jitterbug_amd.vec_env import JitterbugVecEnv
    from oracle import oracle as O
    P = model.default_params()
    n = 64
    rng = np.random.default_rng(2)
    q = np.tile(model.qpos0(), (n, 1))
    quat = rng.normal(size=(n, 4)); q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    v = np.zeros((n, 15)); v[:, :3] = rng.normal(size=(n, 3)) * 0.1; v[:, 3:6] = rng.normal(size=(n, 3)) * 15
    t = np.stack([np.zeros(n), np.zeros(n), rng.uniform(0, 2 * np.pi, n)], 1)
    env = JitterbugVecEnv(n, "move_in_direction")
    env.set_state(q, v, t)
    og, rg = env.observe()
    terms = env.reward_terms()
    differs = 0
    for i in range(n):
        oo = O.observation(P, "move_in_direction", q[i], v[i], t[i])
        np.testing.assert_allclose(og[i], oo, rtol=1e-5, atol=2e-6)
        assert abs(rg[i] - O.reward(P, "move_in_direction", q[i], v[i], t[i])) < 1e-5
        assert abs(terms[i, 2] - O.reward_terms(P, q[i], v[i], t[i])["V"]) < 1e-5
        c, s = np.cos(t[i, 2]), np.sin(t[i, 2])
        differs += abs(oo[16] - (c * v[i, 0] + s * v[i, 1])) > 1e-2
    assert differs > n // 2                                # the COM offset matters at these spin rates
    env.close()


def test_c_abi_row_gather_over_rccl_one_rank():
    """jb_comm_* / jb_gather_rows_device: the rows the step kernel wrote travel through a real RCCL communicator bound by dlopen inside
    the library (no torch.distributed involved).  The test box has one GPU, so the communicator has one rank: rank 0 sends to itself;
    the grouped send/recv path, the run-time binding and the side-stream use are what is exercised.  Run in a subprocess so that a
    communicator never shares a process with torch's own process groups."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import ctypes as C, numpy as np, sys
        sys.path.insert(0, %r)
        import torch
        from jitterbug_amd import _lib, model
        from jitterbug_amd.vec_env import JitterbugVecEnv
        n, task = 777, "move_to_pose"
        D = model.OBS_DIM[task]
        L = _lib.load()
        env = JitterbugVecEnv(n, task, seed=3)
        ref = JitterbugVecEnv(n, task, seed=3)
        env.reset(); ref.reset()
        uid = (C.c_char * 128)()
        _lib.check(L.jb_comm_unique_id(uid))
        _lib.check(L.jb_comm_init(env._h, 1, 0, uid))
        assert L.jb_comm_init(env._h, 1, 0, uid) == -1                     # one communicator per handle
        dev = torch.device("cuda", 0)
        rows = torch.zeros((n, D + 2), device=dev); allrows = torch.zeros((1, n, D + 2), device=dev)
        side = torch.cuda.Stream(device=dev)
        g = torch.Generator().manual_seed(0)
        for t in range(5):
            a = (torch.rand(n, generator=g) * 2 - 1).to(torch.float32)
            ad = a.to(dev)
            env.step_rows_device(ad.data_ptr(), rows.data_ptr()); env.synchronize()
            _lib.check(L.jb_gather_rows_device(env._h, rows.data_ptr(), allrows.data_ptr(), side.cuda_stream, 1))   # on a side stream
            side.synchronize()
            ob, rw, dn, _ = ref.step(a.numpy())
            r = allrows[0].cpu().numpy()
            assert np.array_equal(r[:, :D], ob) and np.array_equal(r[:, D], rw) and np.array_equal(r[:, D + 1] > 0.5, dn.astype(bool)), t
        _lib.check(L.jb_gather_rows_device(env._h, rows.data_ptr(), allrows.data_ptr(), None, 0)); env.synchronize()   # on the handle's stream
        # ABI 5: the action half of the round trip, and the partition check
        bad = np.array([n + 1], dtype=np.int32)
        assert L.jb_comm_set_shards(env._h, bad.ctypes.data) == -1 and b"do not agree" in L.jb_last_error()      # a partition this handle does not fit: refused on the host
        _lib.check(L.jb_comm_set_shards(env._h, np.array([n], dtype=np.int32).ctypes.data))
        a_all = torch.rand((1, n), device=dev) * 2 - 1
        a_loc = torch.zeros(n, device=dev)
        torch.cuda.synchronize()
        _lib.check(L.jb_scatter_actions_device(env._h, a_all.data_ptr(), a_loc.data_ptr(), n, side.cuda_stream, 1)); side.synchronize()
        assert torch.equal(a_loc, a_all[0])
        assert L.jb_scatter_actions_device(env._h, a_all.data_ptr(), a_loc.data_ptr(), n - 1, None, 0) == -1     # a block shorter than the shard
        env.step_rows_device(a_loc.data_ptr(), rows.data_ptr())
        _lib.check(L.jb_gather_rows_device(env._h, rows.data_ptr(), allrows.data_ptr(), None, 0)); env.synchronize()
        ob, rw, dn, _ = ref.step(a_all[0].cpu().numpy())
        r = allrows[0].cpu().numpy()
        assert np.array_equal(r[:, :D], ob) and np.array_equal(r[:, D], rw)
        _lib.check(L.jb_comm_destroy(env._h))
        assert L.jb_gather_rows_device(env._h, rows.data_ptr(), allrows.data_ptr(), None, 0) == -1                    # no communicator any more
        env.close(); ref.close()
        print("CABI_GATHER_OK")
    ''' % root)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CABI_GATHER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_entry_points_leave_the_callers_current_device_alone():
    """with a handle on another GPU than the caller's current one, jb_step / jb_observe / ... must hand the caller's device
    back (torch reads it through hipGetDevice).  Needs two GPUs; on the one-GPU box the guard itself is checked against a stub in
    tests/test_abi_cpu.py and this test only checks that nothing changes."""
    import torch
    from jitterbug_amd.vec_env import JitterbugVecEnv
    ndev = torch.cuda.device_count()
    other = 1 if ndev > 1 else 0
    torch.cuda.set_device(0)
    env = JitterbugVecEnv(8, "move_from_origin", seed=0, device_id=other)
    assert torch.cuda.current_device() == 0
    env.reset(); assert torch.cuda.current_device() == 0
    env.step(np.zeros(8, dtype=np.float32)); assert torch.cuda.current_device() == 0
    env.observe(); env.get_state(); env.counters(); env.randomise_models(seed=1)
    assert torch.cuda.current_device() == 0
    x = torch.zeros(4, device="cuda")
    assert x.device.index == 0
    env.close()
    assert torch.cuda.current_device() == 0
