"""The fused K-step rollout (jb_step_many_device): one launch of K control steps is bit-identical to K single-step launches -
state, packed rows, rewards, counters, in-kernel auto-reset included - for both action sources (an action tape, the heuristic policy
evaluated in the kernel) and for every kernel variant.  Serves the reference's loops benchmarks/evaluate_policy.py:29-33 and the
async split of benchmarks/benchmark.py:146-171 (SURVEY.md 3.3 "one kernel launch per control step or per K control steps")."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def bits(t):
    """float tensor / array -> uint32 bit patterns (NaN-safe exact comparison)"""
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_state(a, b):
    for x, y in zip(a.get_state(), b.get_state()):
        if not np.array_equal(x, y):
            return False
    ca, cb = a.counters(), b.counters()
    return all(np.array_equal(x, y) for x, y in zip(ca, cb))


def make_pair(n, task="move_from_origin", **kw):
    from jitterbug_amd.vec_env import JitterbugVecEnv
    return JitterbugVecEnv(n, task, seed=7, **kw), JitterbugVecEnv(n, task, seed=7, **kw)


CASES = [
    # (label, n_envs, kwargs, randomise)
    ("n4096-ordinary", 4096, dict(), False),
    ("n8192-lean", 8192, dict(flags=2), False),
    ("n256-per-env-models-pair", 256, dict(), True),
    ("n1024-lean-pair", 1024, dict(flags=2), True),
    ("n37-ragged-two-envs-per-wave", 37, dict(envs_per_wave=2), False),
]


@pytest.mark.parametrize("label,n,kw,randomise", CASES, ids=[c[0] for c in CASES])
def test_action_tape_rollout_is_bit_identical_to_single_steps(label, n, kw, randomise):
    torch = _torch()
    dev = torch.device("cuda", 0)
    K = 45
    # a 0.3 s time limit = 30 control steps: the rollout crosses an in-kernel auto-reset (new episode, new Philox draws, new target)
    a, b = make_pair(n, "move_to_pose", time_limit=0.3, **kw)
    try:
        if randomise:
            a.randomise_models(seed=5, return_params=False)
            b.randomise_models(seed=5, return_params=False)
        assert a.kernel_variant == b.kernel_variant == {"n4096-ordinary": "ordinary", "n8192-lean": "lean", "n256-per-env-models-pair": "pair",
                                                        "n1024-lean-pair": "lean_pair", "n37-ragged-two-envs-per-wave": "ordinary"}[label]
        D = a.obs_dim
        g = torch.Generator(device=dev); g.manual_seed(3)
        tape = torch.rand((K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        tape[:, : n // 3] = 1.0          # a third of the robots flat out: some tip over (all-geom path, spread sweeps)
        a.reset_device(); b.reset_device()
        rows_a = torch.full((K, n, D + 2), float("nan"), device=dev)
        rows_b = torch.full((K, n, D + 2), float("nan"), device=dev)
        for k in range(K):
            a.step_rows_device(tape[k].data_ptr(), rows_a[k].data_ptr())
        b.step_many_device(K, tape.data_ptr(), rows_ptr=rows_b.data_ptr())
        a.synchronize(); b.synchronize()
        assert np.isfinite(rows_a.cpu().numpy()).all()
        assert np.array_equal(bits(rows_a), bits(rows_b)), "packed rows of the fused rollout differ from the single steps'"
        assert same_state(a, b)
        done = rows_a[:, :, D + 1].cpu().numpy()
        assert done[29].all() and done.sum() == n, "every env finished exactly one episode inside the rollout"
        # two fused halves = one fused whole (state carried through HBM in between), and the separate-output layout agrees with the rows
        a.reset_device(); b.reset_device()
        rew = torch.zeros((K, n), device=dev); obs_last = torch.zeros((n, D), device=dev); done_last = torch.zeros((n,), device=dev, dtype=torch.uint8)
        a.step_many_device(20, tape[:20].data_ptr(), rewards_ptr=rew[:20].data_ptr())
        a.step_many_device(K - 20, tape[20:].data_ptr(), rewards_ptr=rew[20:].data_ptr(), obs_last_ptr=obs_last.data_ptr(), done_last_ptr=done_last.data_ptr())
        b.step_many_device(K, tape.data_ptr(), rows_ptr=rows_b.data_ptr())
        a.synchronize(); b.synchronize()
        assert same_state(a, b)
        assert np.array_equal(bits(rew), bits(rows_b[:, :, D])) and np.array_equal(bits(obs_last), bits(rows_b[K - 1, :, :D]))
        assert np.array_equal(done_last.cpu().numpy(), rows_b[K - 1, :, D + 1].cpu().numpy().astype(np.uint8))
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("task", ["move_from_origin", "face_direction", "move_in_direction", "move_to_position", "move_to_pose"])
def test_in_kernel_policy_rollout_is_bit_identical_to_policy_then_step(task):
    """actions = NULL: the heuristic policy is evaluated inside the step kernel on the observation the lanes just produced.  Same bits
    as the chain jb_policy_device -> jb_step_device the reference's evaluate_policy loop maps to (the task layer is compiled without
    multiply-add contraction, so an observation has the same bits whichever kernel computed it)."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, K = 512, 60
    a, b = make_pair(n, task, time_limit=0.4)
    try:
        D = a.obs_dim
        for e in (a, b):
            e.set_policy_params(kick_angle=0.6, speed=0.8, angle_threshold=0.3)
        obs = torch.zeros((n, D), device=dev); act = torch.zeros((n,), device=dev)
        rew_a = torch.zeros((K, n), device=dev); rew_b = torch.zeros((K, n), device=dev)
        done = torch.zeros((n,), device=dev, dtype=torch.uint8)
        a.reset_device(None, obs.data_ptr()); b.reset_device()
        for k in range(K):
            a.policy_device(obs.data_ptr(), act.data_ptr())
            a.step_device(act.data_ptr(), obs.data_ptr(), rew_a[k].data_ptr(), done.data_ptr())
        obs_b = torch.zeros((n, D), device=dev); done_b = torch.zeros((n,), device=dev, dtype=torch.uint8)
        b.step_many_device(K, None, rewards_ptr=rew_b.data_ptr(), obs_last_ptr=obs_b.data_ptr(), done_last_ptr=done_b.data_ptr())
        a.synchronize(); b.synchronize()
        assert np.array_equal(bits(rew_a), bits(rew_b))
        assert np.array_equal(bits(obs), bits(obs_b)) and np.array_equal(done.cpu().numpy(), done_b.cpu().numpy())
        assert same_state(a, b)
        assert float(rew_a.abs().sum()) > 0
        # ... and the rollout entry point of round 2 (observations in / out) is the same launch
        a.reset_device(None, obs.data_ptr()); b.reset_device(None, obs_b.data_ptr())
        for k in range(10):
            a.policy_device(obs.data_ptr(), act.data_ptr())
            a.step_device(act.data_ptr(), obs.data_ptr(), rew_a[k].data_ptr(), done.data_ptr())
        b.rollout_policy_device(10, obs_b.data_ptr(), rew_b.data_ptr(), done_b.data_ptr())
        a.synchronize(); b.synchronize()
        assert np.array_equal(bits(rew_a[:10]), bits(rew_b[:10])) and np.array_equal(bits(obs), bits(obs_b)) and same_state(a, b)
    finally:
        a.close(); b.close()


def test_full_episode_rollout_n4096_bit_identical_and_wave_clocks():
    """BASELINE configs[2] at full size: 1000 steps + the auto-reset in one launch = 1000 launches; the per-wave clocks of the fused
    launch show what it removes (every launch of the step-by-step path lasts as long as ITS slowest wave)."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    n, K = 4096, 1000
    a, b = make_pair(n)
    try:
        D = a.obs_dim
        g = torch.Generator(device=dev); g.manual_seed(11)
        tape = torch.rand((K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        obs = torch.zeros((n, D), device=dev); rew_a = torch.zeros((K, n), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
        obs_b = torch.zeros((n, D), device=dev); rew_b = torch.zeros((K, n), device=dev); done_b = torch.zeros((n,), device=dev, dtype=torch.uint8)
        a.reset_device(); b.reset_device()
        for k in range(K):
            a.step_device(tape[k].data_ptr(), obs.data_ptr(), rew_a[k].data_ptr(), done.data_ptr())
        b.step_many_device(K, tape.data_ptr(), rewards_ptr=rew_b.data_ptr(), obs_last_ptr=obs_b.data_ptr(), done_last_ptr=done_b.data_ptr())
        a.synchronize(); b.synchronize()
        assert np.array_equal(bits(rew_a), bits(rew_b)) and np.array_equal(bits(obs), bits(obs_b))
        assert done.cpu().numpy().all() and done_b.cpu().numpy().all()
        assert same_state(a, b)
        sc, ep, cap = b.counters()
        assert (sc == 0).all() and (ep == ep[0]).all() and float(cap.max()) < 1000.0
        wc = b.wave_clocks()
        assert wc.shape == (n // b.envs_per_wave,) and (wc > 0).all()
        print("fused 1000-step launch at N=4096: mean wave %.1f ms, slowest wave %.1f ms (ratio %.3f)" % (1e3 * wc.mean(), 1e3 * wc.max(), wc.mean() / wc.max()))
        assert wc.max() < 5.0
    finally:
        a.close(); b.close()


def test_step_many_argument_checks_and_variant_refusals():
    from jitterbug_amd import _lib
    from jitterbug_amd.vec_env import JitterbugVecEnv
    torch = _torch()
    dev = torch.device("cuda", 0)
    env = JitterbugVecEnv(64, "move_from_origin", seed=1)
    try:
        D = env.obs_dim
        rows = torch.zeros((2, 64, D + 2), device=dev); rew = torch.zeros((2, 64), device=dev); tape = torch.zeros((2, 64), device=dev)
        q0 = env.get_state()[0].copy()
        env.step_many_device(0, tape.data_ptr(), rows_ptr=rows.data_ptr())          # K = 0: nothing happens
        assert np.array_equal(env.get_state()[0], q0)
        with pytest.raises(_lib.JitterbugHipError):
            env.step_many_device(2, tape.data_ptr(), rows_ptr=rows.data_ptr(), rewards_ptr=rew.data_ptr())      # one output layout per launch
        with pytest.raises(_lib.JitterbugHipError):
            env.step_many_device(-1, tape.data_ptr())
        env.step_many_device(2, tape.data_ptr())            # no outputs at all is fine: the state advances
        assert not np.array_equal(env.get_state()[0], q0)
    finally:
        env.close()
    # JB_FLAG_LEAN where no LEAN instantiation exists is refused, never silently run as the ordinary kernel (VERDICT r3 / ADVICE r3)
    with pytest.raises(_lib.JitterbugHipError, match="JB_FLAG_LEAN"):
        JitterbugVecEnv(64, "move_from_origin", flags=_lib.FLAG_LEAN | _lib.FLAG_PAIR)            # shared model + forced pair contact
    env = JitterbugVecEnv(64, "move_from_origin", flags=_lib.FLAG_LEAN, envs_per_wave=2)
    try:
        assert env.kernel_variant == "lean" and env.envs_per_wave == 2
        with pytest.raises(_lib.JitterbugHipError, match="JB_FLAG_LEAN"):
            env.randomise_models(seed=1)               # one model per env needs 4 envs per wave in the LEAN layout
        assert env.kernel_variant == "lean"            # the handle kept the model it had
        env.step(np.zeros(64, np.float32))
    finally:
        env.close()


@pytest.mark.parametrize("layout,n,lean", [("one-wave-two-rounds", 8192, False), ("two-waves-folded", 8192, True), ("two-waves-folded-ragged", 6004, True)],
                         ids=["one-wave-two-rounds", "two-waves-folded", "two-waves-folded-ragged"])
@pytest.mark.parametrize("randomise", [False, True], ids=["nominal", "per-env-models"])
def test_longest_first_wave_order_does_not_change_a_single_bit(randomise, layout, n, lean):
    """A batch with more waves than the device holds at once (8192 envs = 2048 four-env waves on 1024 SIMDs) launches its waves
    longest-first, by the wave times the previous launch measured (jb_wave_order_kernel); the two-waves-per-SIMD kernels, while the batch
    still fits the device at once, FOLD that order so that the longest wave shares its SIMD with the shortest (workgroups b and b + 1024
    land on one SIMD), 6004 envs: 477 waves keep a SIMD to themselves; JB_FLAG_NO_REORDER keeps the plain order.
    Which workgroup steps which envs must never show in the results: single steps and a fused rollout, bit for bit."""
    from jitterbug_amd import _lib
    from jitterbug_amd.vec_env import JitterbugVecEnv
    torch = _torch()
    dev = torch.device("cuda", 0)
    K = 24
    a = JitterbugVecEnv(n, "move_to_pose", seed=13, time_limit=0.2, flags=_lib.FLAG_LEAN if lean else 0)
    b = JitterbugVecEnv(n, "move_to_pose", seed=13, time_limit=0.2, flags=_lib.FLAG_NO_REORDER | (_lib.FLAG_LEAN if lean else 0))
    try:
        if randomise:
            a.randomise_models(seed=2, return_params=False); b.randomise_models(seed=2, return_params=False)
        D = a.obs_dim
        g = torch.Generator(device=dev); g.manual_seed(8)
        tape = torch.rand((2 * K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        tape[:, ::7] = 1.0
        rows_a = torch.zeros((2 * K, n, D + 2), device=dev); rows_b = torch.zeros((2 * K, n, D + 2), device=dev)
        a.reset_device(); b.reset_device()
        for k in range(K):
            a.step_rows_device(tape[k].data_ptr(), rows_a[k].data_ptr())
            b.step_rows_device(tape[k].data_ptr(), rows_b[k].data_ptr())
        a.step_many_device(K, tape[K:].data_ptr(), rows_ptr=rows_a[K:].data_ptr())
        b.step_many_device(K, tape[K:].data_ptr(), rows_ptr=rows_b[K:].data_ptr())
        a.synchronize(); b.synchronize()
        assert np.array_equal(bits(rows_a), bits(rows_b)) and same_state(a, b)
        wa, wb = a.wave_clocks(), b.wave_clocks()
        assert wa.shape == wb.shape == (n // 4,) and (wa > 0).all() and (wb > 0).all()
        assert a.kernel_variant == b.kernel_variant == (("lean_pair" if randomise else "lean") if lean else ("pair" if randomise else "ordinary"))
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("label,n,task,randomise", [("configs3-32768-move_to_pose", 32768, "move_to_pose", False), ("configs4-65536-randomised", 65536, "move_from_origin", True)],
                         ids=["configs3", "configs4"])
def test_fused_rollout_at_the_full_baseline_sizes(label, n, task, randomise):
    """BASELINE configs[3] (32 768 envs, move_to_pose) and configs[4] (65 536 envs, one randomised model each) whole on one GPU, with the
    product's own variant selection (two waves per SIMD; configs[4]: LEAN + PAIR with the per-substep table overlay), waves launched
    longest-first: one fused K-step launch = K single steps bit for bit, every env stays physical, an episode ends inside."""
    from jitterbug_amd.vec_env import JitterbugVecEnv
    torch = _torch()
    dev = torch.device("cuda", 0)
    K = 24
    kw = dict(seed=5, variant="auto", per_env_model=randomise, time_limit=0.2)
    a, b = JitterbugVecEnv(n, task, **kw), JitterbugVecEnv(n, task, **kw)
    try:
        if randomise:
            a.randomise_models(seed=9, return_params=False); b.randomise_models(seed=9, return_params=False)
        assert a.kernel_variant == ("lean_pair" if randomise else "lean")
        D = a.obs_dim
        g = torch.Generator(device=dev); g.manual_seed(2)
        tape = torch.rand((K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        rows_a = torch.zeros((K, n, D + 2), device=dev); rows_b = torch.zeros((K, n, D + 2), device=dev)
        a.reset_device(); b.reset_device()
        for k in range(K):
            a.step_rows_device(tape[k].data_ptr(), rows_a[k].data_ptr())
        b.step_many_device(K, tape.data_ptr(), rows_ptr=rows_b.data_ptr())
        a.synchronize(); b.synchronize()
        assert np.array_equal(bits(rows_a), bits(rows_b)) and same_state(a, b)
        r = rows_b.cpu().numpy()
        assert np.isfinite(r).all() and r[19, :, D + 1].all() and r[:, :, D + 1].sum() == n
        z = (r[-1, :, 2] + 1) / 20
        assert z.min() > 0.005 and z.max() < 0.08 and np.abs(np.linalg.norm(r[-1, :, 3:7], axis=1) - 1).max() < 1e-5
        sc, ep, cap = b.counters()
        assert cap.sum() == 0
    finally:
        a.close(); b.close()


def test_host_buffer_rollout_equals_step_by_step():
    """JitterbugVecEnv.rollout (jb_step_many: numpy in, numpy out - no torch needed) = the same steps through step(), bit for bit, for an
    action tape and for the in-kernel policy; the reference's evaluate_policy loop (benchmarks/evaluate_policy.py:29-33) in one call."""
    from jitterbug_amd.vec_env import JitterbugVecEnv
    n, K = 300, 40
    a, b = JitterbugVecEnv(n, "move_to_position", seed=2, time_limit=0.25), JitterbugVecEnv(n, "move_to_position", seed=2, time_limit=0.25)
    try:
        rng = np.random.default_rng(0)
        tape = rng.uniform(-1, 1, size=(K, n)).astype(np.float32)
        a.reset(); b.reset()
        ob, rw, dn = b.rollout(K, tape)
        for k in range(K):
            o1, r1, d1, _ = a.step(tape[k])
            assert np.array_equal(o1, ob[k]) and np.array_equal(r1, rw[k]) and np.array_equal(d1, dn[k]), k
        assert dn.sum() == n and same_state(a, b)
        oa = a.observe()[0]
        ob2, rw2, dn2 = b.rollout(K)                     # the heuristic policy acts inside the kernel
        for k in range(K):
            oa, r1, d1, _ = a.step(a.policy(oa))
            assert np.array_equal(oa, ob2[k]) and np.array_equal(r1, rw2[k]) and np.array_equal(d1, dn2[k]), k
        assert same_state(a, b)
    finally:
        a.close(); b.close()
