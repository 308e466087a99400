"""The simulator source the HIP kernel runs (jitterbug_amd/csrc/jb_sim.hpp), compiled for the host with the 4 lanes of
an env emulated by jb::Quad<T> (tests/host_harness.cpp), against the independent oracle formulation:
  * fp64 instantiation: agreement to round-off proves the star-topology/Schur/Newton scheme EQUALS the oracle's
    world-frame projection dynamics + dense contact solve (two formulations, one answer);
  * fp32 instantiation: what the fp64 -> fp32 narrowing alone costs, without a GPU."""
import ctypes as C

import numpy as np
import pytest

from jitterbug_amd import model
from oracle import oracle as O


@pytest.fixture(scope="module")
def hstep():
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P0 = model.default_params()

    def step(q, v, u, nsub=50, contacts=1, maxn=20, f32=0, P=None):
        P = np.ascontiguousarray(P0 if P is None else P, dtype=np.float64)
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        rc = lib.jbh_step(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), float(u), nsub, contacts, maxn, 1, f32,
                          fail.ctypes.data_as(dp))
        assert rc == 0, rc
        return q, v, fail[0]
    return step


def test_smooth_dynamics_fp64_equals_oracle(hstep, params):
    rng = np.random.default_rng(0)
    o = O.default_opts(contacts=0)
    for trial in range(5):
        q = model.qpos0(params)
        q[3:7] = rng.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
        q[7:15] = rng.normal(size=8) * 0.05; q[15] = rng.uniform(-3, 3)
        v = rng.normal(size=15) * np.array([.1] * 3 + [2] * 3 + [3] * 8 + [50])
        qo, vo = O.step_physics(params, q, v, 0.4, 50, o)
        qh, vh, _ = hstep(q, v, 0.4, 50, contacts=0)
        np.testing.assert_allclose(qh, qo, rtol=0, atol=1e-13)
        np.testing.assert_allclose(vh, vo, rtol=1e-12, atol=1e-12)


def test_contact_rollout_fp64_equals_oracle_and_fp32_is_close(hstep, params):
    n = 8
    env = O.OracleEnv(n, "move_from_origin", params, seed=0)
    env.reset()
    rng = np.random.default_rng(1)
    worst64, errs32 = 0.0, []
    for t in range(40):
        a = rng.uniform(-1, 1, size=n)
        q0, v0, _ = env.get_state()
        env.step(a, auto_reset=False)
        q1, v1, _ = env.get_state()
        for i in range(n):
            qh, vh, cap = hstep(q0[i], v0[i], a[i])
            assert cap == 0
            worst64 = max(worst64, np.abs(qh - q1[i]).max(), np.abs(vh - v1[i]).max() / 100)
            qf, vf, _ = hstep(q0[i], v0[i], a[i], f32=1)
            errs32.append(max(np.abs(qf[:7] - q1[i][:7]).max(), (np.abs(vf[:6] - v1[i][:6]) / np.array([1, 1, 1, 35, 35, 35])).max()))
    assert worst64 < 1e-11
    errs32 = np.array(errs32)
    assert np.median(errs32) < 1e-6 and np.quantile(errs32, 0.95) < 1e-5


def test_every_geom_type_fp64_equals_oracle(hstep, params):
    """Random orientations pushed into the floor: upper legs, knee tips, both boxes, cylinders, ellipsoids, motor-body geoms."""
    rng = np.random.default_rng(5)
    seen = set()
    for trial in range(60):
        q = model.qpos0(params)
        q[3:7] = rng.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
        q[7:15] = rng.normal(size=8) * 0.03; q[15] = rng.uniform(-3, 3)
        v = rng.normal(size=15) * np.array([.05] * 3 + [1] * 3 + [1] * 8 + [20])
        lo, hi = -0.1, 0.2
        for _ in range(30):
            mid = 0.5 * (lo + hi); q[2] = mid
            if O.forward_debug(params, q, v, 0.0)["ncon"] > 0:
                lo = mid
            else:
                hi = mid
        q[2] = lo - rng.uniform(0.0002, 0.002) - (0.03 if trial % 2 else 0.0)
        u = rng.uniform(-1, 1)
        seen |= set(O.forward_debug(params, q, v, u)["con_geom"].tolist())
        qo, vo = O.step_physics(params, q, v, u, 2)
        qh, vh, cap = hstep(q, v, u, 2)
        assert cap == 0
        assert np.abs(qh - qo).max() < 1e-12 and (np.abs(vh - vo) / (1 + np.abs(vo))).max() < 1e-10
    assert seen == set(range(22))


def test_randomised_models_fp64_equals_oracle(hstep):
    """Perturbed geometry (shoulder and knee axes no longer parallel, shifted anchors, moved motor axis): the star-topology
    scheme must still equal the oracle."""
    from jitterbug_amd import augmented_jitterbug as aj
    Ps = aj.augmented_params(6, seed=11)
    rng = np.random.default_rng(2)
    for P in Ps:
        env = O.OracleEnv(1, "move_from_origin", P, seed=0)
        env.reset()
        for t in range(6):
            a = rng.uniform(-1, 1, size=1)
            q0, v0, _ = env.get_state()
            env.step(a, auto_reset=False)
            q1, v1, _ = env.get_state()
            qh, vh, cap = hstep(q0[0], v0[0], a[0], P=P)
            assert cap == 0
            assert np.abs(qh - q1[0]).max() < 1e-11 and (np.abs(vh - v1[0]) / (1 + np.abs(v1[0]))).max() < 1e-10


# ----------------------------------------------------------------------------------------------- helper groups on the host
@pytest.fixture(scope="module")
def gstep():
    """jbh_step_groups: the kernel source with FOUR lane groups (one main + three helper groups, one host thread each, exchanging
    through mailboxes where the device uses permlane swaps / readfirstlane): group_sum, row_transpose_sum, the slot->group plan, the
    rank-one pass on rows built by other groups and the broadcast loop decisions run in fp64."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_groups.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P0 = model.default_params()

    def step(q, v, u, nsub=50, groups=4, f32=0, rank_one=1, P=None, maxn=20):
        P = np.ascontiguousarray(P0 if P is None else P, dtype=np.float64)
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        rc = lib.jbh_step_groups(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), float(u), nsub, 1, maxn, f32, groups, rank_one, fail.ctypes.data_as(dp))
        assert rc == 0, rc
        return q, v, fail[0]
    return step


def _contact_states(params, n, tipped):
    """states from oracle rollouts: ordinary walking, or robots driven flat out until some lie on their legs (all-geom path)"""
    env = O.OracleEnv(n, "move_from_origin", params, seed=5, step_limit=10 ** 9)
    env.reset()
    rng = np.random.default_rng(2)
    for t in range(260 if tipped else 30):
        env.step(np.ones(n) if tipped else rng.uniform(-1, 1, size=n), auto_reset=False)
    return env


def test_helper_groups_fp64_equal_single_group_and_oracle(gstep, params):
    worst_groups = worst_oracle = 0.0
    n_multi = 0
    for tipped in (False, True):
        env = _contact_states(params, 12, tipped)
        rng = np.random.default_rng(7)
        for t in range(6):
            a = rng.uniform(-1, 1, size=12)
            q0, v0, _ = env.get_state()
            env.step(a, auto_reset=False)
            q1, v1, _ = env.get_state()
            for i in range(12):
                d = O.forward_debug(params, q0[i], v0[i], a[i])
                n_multi += d["ncon"] >= 2                                   # >= 2 live slots: the groups really share the work
                qg, vg, cap = gstep(q0[i], v0[i], a[i], groups=4)
                qs, vs, _ = gstep(q0[i], v0[i], a[i], groups=1)
                assert cap == 0
                worst_groups = max(worst_groups, np.abs(qg - qs).max(), (np.abs(vg - vs) / (1 + np.abs(vs))).max())
                worst_oracle = max(worst_oracle, np.abs(qg - q1[i]).max(), (np.abs(vg - v1[i]) / (1 + np.abs(v1[i]))).max())
        if tipped:
            q, _, _ = env.get_state()
            assert ((1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)) < 0.5).any()    # some robots really lie on the floor
    print("4 groups vs 1 group: %.2e   4 groups vs oracle: %.2e   (%d multi-contact starts)" % (worst_groups, worst_oracle, n_multi))
    assert n_multi > 40 and worst_groups < 1e-11 and worst_oracle < 1e-10


def test_every_geom_type_with_lane_groups_fp64_equals_oracle(gstep, params):
    """test_every_geom_type_fp64_equals_oracle's random orientations pushed into the floor, with TWO and FOUR lane groups: the body slots
    (root box vertices, screws, motor-body geoms: slots 10-27) are dealt to different groups, whose accumulators and active-set records
    are combined across groups every pass - up to fourteen contacts at once, four substeps, against the oracle."""
    rng = np.random.default_rng(5)
    seen, most, worst = set(), 0, 0.0
    for trial in range(70):
        q = model.qpos0(params)
        q[3:7] = rng.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
        q[7:15] = rng.normal(size=8) * 0.03; q[15] = rng.uniform(-3, 3)
        v = rng.normal(size=15) * np.array([.05] * 3 + [1] * 3 + [1] * 8 + [20])
        lo, hi = -0.1, 0.2
        for _ in range(30):
            mid = 0.5 * (lo + hi); q[2] = mid
            if O.forward_debug(params, q, v, 0.0)["ncon"] > 0:
                lo = mid
            else:
                hi = mid
        q[2] = lo - rng.uniform(0.0002, 0.004)
        u = rng.uniform(-1, 1)
        d = O.forward_debug(params, q, v, u)
        seen |= set(d["con_geom"][:d["ncon"]].tolist()); most = max(most, d["ncon"])
        qo, vo = O.step_physics(params, q, v, u, 4)
        for groups in (2, 4):
            qh, vh, cap = gstep(q, v, u, 4, groups=groups, maxn=12)
            assert cap == 0
            worst = max(worst, np.abs(qh - qo).max(), (np.abs(vh - vo) / (1 + np.abs(vo))).max())
    print("random orientations on the floor, 2 and 4 lane groups vs oracle: %.2e (geoms seen %d, most contacts at once %d)" % (worst, len(seen), most))
    assert len(seen) >= 14 and most >= 7
    assert worst < 1e-10


def test_rank_one_passes_fp64_equal_full_passes_with_groups(gstep, params):
    """Sherman-Morrison on the kept factorisation, reading contact rows that OTHER groups built (the path of round 1's
    uninitialised-row bug), and the rank-one update / downdate of that factorisation after every pass (chains of single-edge
    changes, several edges per check), against full sweeps + refactorisation: same minimiser to round-off."""
    worst = 0.0
    for tipped in (False, True):          # tipped: legs lying on the floor, 8+ contacts per leg, several edges changing per check (chained passes)
        env = _contact_states(params, 16, tipped=tipped)
        rng = np.random.default_rng(9)
        for t in range(4):
            a = rng.uniform(-1, 1, size=16)
            q0, v0, _ = env.get_state()
            env.step(a, auto_reset=False)
            for i in range(16):
                qa, va, ca = gstep(q0[i], v0[i], a[i], groups=4, rank_one=1)
                qb, vb, _ = gstep(q0[i], v0[i], a[i], groups=4, rank_one=0)
                assert ca == 0
                worst = max(worst, np.abs(qa - qb).max(), (np.abs(va - vb) / (1 + np.abs(vb))).max())
    assert worst < 1e-10, worst


def test_helper_groups_fp32_agree_with_single_group_within_rounding(gstep, params):
    env = _contact_states(params, 8, tipped=False)
    q0, v0, _ = env.get_state()
    errs = []
    for i in range(8):
        qg, vg, _ = gstep(q0[i], v0[i], 0.3, groups=4, f32=1)
        qs, vs, _ = gstep(q0[i], v0[i], 0.3, groups=1, f32=1)
        errs.append(max(np.abs(qg[:7] - qs[:7]).max(), np.abs(vg[:6] - vs[:6]).max() / 35))
    assert np.median(errs) < 1e-6


def test_lean_variant_is_bit_identical_on_the_host(params):
    """JB_FLAG_LEAN (the 256-register kernel variant: constants read from LDS, lane state / joint-space system / kept factorisation parked
    in the scratch between phases) runs the same arithmetic in the same order: fp64 AND fp32 host builds reproduce the ordinary variant
    bit for bit, with one and with four lane groups, walking and tipped over."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    for f in (lib.jbh_step_groups, lib.jbh_step_lean):
        f.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P = np.ascontiguousarray(params)

    def run(fn, q, v, groups, f32):
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        assert fn(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), 0.3, 50, 1, 20, f32, groups, 1, fail.ctypes.data_as(dp)) == 0
        return q, v

    lib.jbh_set_aux(0)          # (the aux bodies of the ordinary kernel sum the motor / root terms in another order: compared separately below)
    try:
        for tipped in (False, True):
            env = _contact_states(params, 6, tipped)
            q0, v0, _ = env.get_state()
            for i in range(6):
                for groups in (1, 4):
                    for f32 in (0, 1):
                        a, b = run(lib.jbh_step_groups, q0[i], v0[i], groups, f32), run(lib.jbh_step_lean, q0[i], v0[i], groups, f32)
                        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (tipped, i, groups, f32)
    finally:
        lib.jbh_set_aux(1)


def test_aux_bodies_on_the_helper_groups_equal_the_replicated_path(params):
    """SimOpts::aux (the ordinary device kernel with four lane groups): lane groups 2 / 3 run phase A on the motor body and the root body's own
    mass as two more "legs" - the same instruction stream as the legs - and hand the sums over with cross-lane swaps, instead of every leg
    lane repeating that work.  Same physics, another order of summation: fp64 agrees with the replicated path to round-off (and with the
    oracle: test_helper_groups_fp64_equal_single_group_and_oracle runs with it on), fp32 to its own rounding; walking and tipped over."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_groups.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P = np.ascontiguousarray(params)

    def run(q, v, f32, aux):
        lib.jbh_set_aux(aux)
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        try:
            assert lib.jbh_step_groups(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), 0.6, 50, 1, 20, f32, 4, 1, fail.ctypes.data_as(dp)) == 0
        finally:
            lib.jbh_set_aux(1)
        assert fail[0] == 0
        return q, v

    w64, w32 = 0.0, []
    for tipped in (False, True):
        env = _contact_states(params, 8, tipped)
        q0, v0, _ = env.get_state()
        for i in range(8):
            a, b = run(q0[i], v0[i], 0, 1), run(q0[i], v0[i], 0, 0)
            w64 = max(w64, np.abs(a[0] - b[0]).max(), (np.abs(a[1] - b[1]) / (1 + np.abs(b[1]))).max())
            a, b = run(q0[i], v0[i], 1, 1), run(q0[i], v0[i], 1, 0)
            w32.append(max(np.abs(a[0][:7] - b[0][:7]).max(), np.abs(a[1][:6] - b[1][:6]).max() / 35))
    print("aux bodies vs replicated path: fp64 %.2e, fp32 median %.2e" % (w64, np.median(w32)))
    assert w64 < 1e-11 and np.median(w32) < 2e-6


def test_replica_group_offload_is_bit_identical_on_the_host(params):
    """SimOpts::offload (the one-wave-per-SIMD kernels): lane group 1 replicates the main lanes, factorises M + h diag(b) with the instruction
    stream of the first Newton solve and does the final pass as a substitution.  Same arithmetic on the same numbers: fp64 and fp32 host
    builds reproduce the path without it bit for bit (walking, tipped over, the pair-contact variant)."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    for f in (lib.jbh_step_groups, lib.jbh_step_pair):
        f.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P = np.ascontiguousarray(params)

    def run(fn, q, v, f32, offload, u):
        lib.jbh_set_offload(offload)
        lib.jbh_set_aux(0)          # (the aux bodies ride on the offload layout and reorder the motor / root sums: they have their own test)
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        try:
            assert fn(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), u, 50, 1, 20, f32, 4, 1, fail.ctypes.data_as(dp)) == 0
        finally:
            lib.jbh_set_offload(1)
            lib.jbh_set_aux(1)
        return q, v

    for tipped in (False, True):
        env = _contact_states(params, 6, tipped)
        q0, v0, _ = env.get_state()
        for i in range(6):
            for f32 in (0, 1):
                for fn in (lib.jbh_step_groups, lib.jbh_step_pair):
                    a, b = run(fn, q0[i], v0[i], f32, 1, 0.3), run(fn, q0[i], v0[i], f32, 0, 0.3)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (tipped, i, f32)
    # a robot in free fall (no contact anywhere: the final pass factorises on the spot, in both groups)
    q = q0[0].copy(); q[2] += 0.05
    a, b = run(lib.jbh_step_groups, q, v0[0], 0, 1, -0.5), run(lib.jbh_step_groups, q, v0[0], 0, 0, -0.5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _lying_states(params, n):
    """robots dropped at random attitudes and left to settle: they end up lying on legs and body, 8 contacts or so each"""
    env = O.OracleEnv(n, "move_from_origin", params, seed=5, step_limit=10 ** 9)
    env.reset()
    q, v, _ = env.get_state()
    rng = np.random.default_rng(3)
    quat = rng.normal(size=(n, 4))
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    q[:, 2] = 0.06
    v[:] = 0
    env.set_state(q, v)
    for _ in range(150):
        env.step(np.zeros(n), auto_reset=False)
    return env


def test_spread_sweeps_fp64_equal_oracle_on_lying_robots(params):
    """Spread contact sweeps (a plan with more than one round: the lanes of idle legs adopt contacts of a leg that has several, leg-block sums
    and active-set records routed to the owning lane): the fp64 host build with four lane groups reproduces the oracle on robots lying on
    the floor, with the spread mode on and off; the two differ in bits (another order of summation), so the spread path really ran."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_groups.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P = np.ascontiguousarray(params)

    def run(q, v, u, spread, rank_one=1, groups=4):
        lib.jbh_set_spread(spread)
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        try:
            assert lib.jbh_step_groups(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), float(u), 50, 1, 20, 0, groups, rank_one, fail.ctypes.data_as(dp)) == 0
        finally:
            lib.jbh_set_spread(1)
        assert fail[0] == 0
        return q, v

    n = 16
    env = _lying_states(params, n)
    ncon = []
    worst = {1: 0.0, 0: 0.0}
    n_diff = 0
    rng = np.random.default_rng(11)
    for t in range(3):
        q0, v0, _ = env.get_state()
        u = rng.uniform(-1, 1, size=n)
        ncon += [O.forward_debug(params, q0[i], v0[i], u[i])["ncon"] for i in range(n)]
        env.step(u, auto_reset=False)
        q1, v1, _ = env.get_state()
        for i in range(n):
            res = {sp: run(q0[i], v0[i], u[i], sp) for sp in (1, 0)}
            for sp, (q, v) in res.items():
                worst[sp] = max(worst[sp], np.abs(q - q1[i]).max(), (np.abs(v - v1[i]) / (1 + np.abs(v1[i]))).max())
            n_diff += not (np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1]))
            if t == 0:          # the two-group layout of the 8-envs-per-wave kernel (five leg slots per group: up to four spread rounds' worth of overflow)
                q, v = run(q0[i], v0[i], u[i], 1, groups=2)
                worst[1] = max(worst[1], np.abs(q - q1[i]).max(), (np.abs(v - v1[i]) / (1 + np.abs(v1[i]))).max())
    print("lying robots: contacts %d-%d, spread vs oracle %.2e, ordinary vs oracle %.2e, %d of %d differ in bits" % (min(ncon), max(ncon), worst[1], worst[0], n_diff, 3 * n))
    assert max(ncon) >= 8 and worst[1] < 1e-10 and worst[0] < 1e-10 and n_diff >= 5


def test_pair_contact_kernel_source_fp64_equals_oracle():
    """The PAIR instantiation of the substep (closed-form narrow phase of the mass ellipsoid against the upper-leg cylinders, the pair's own
    contact frame, rows without root columns, the shoulder - motor cross term folded into the motor branch of the star solve, the 53rd
    value of the group reduction) on robots whose mass really hits a front leg: fp64 = the oracle's dense formulation to round-off, with
    one lane group and with four; the fp32 build is within single-precision rounding of it."""
    import tests.build_harness as bh
    from jitterbug_amd import augmented_jitterbug as aj
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_pair.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]

    def step(P, q, v, u, f32=0, groups=1):
        P = np.ascontiguousarray(P); q = q.copy(); v = v.copy(); fail = np.zeros(1)
        assert lib.jbh_step_pair(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), float(u), 50, 1, 20, f32, groups, 1, fail.ctypes.data_as(dp)) == 0
        assert fail[0] == 0
        return q, v

    Ps = aj.augmented_params(600, seed=123)
    bad = np.nonzero(O.mass_sweep_clearance(Ps, 72) <= 1e-9)[0]
    o = O.default_opts(pair_contacts=1)
    worst, n_pair, e32 = 0.0, 0, []
    for i in bad[:5]:
        P = Ps[i]
        q, v = model.qpos0(P), np.zeros(model.NV)
        rng = np.random.default_rng(int(i))
        for t in range(24):
            u = rng.uniform(-1, 1)
            n_pair += int((O.forward_debug(P, q, v, u, o)["con_geom"] >= 22).any())
            q1, v1 = O.step_physics(P, q, v, u, 50, o)
            for g in (1, 4):
                qh, vh = step(P, q, v, u, groups=g)
                worst = max(worst, np.abs(qh - q1).max(), (np.abs(vh - v1) / (1 + np.abs(v1))).max())
            qf, vf = step(P, q, v, u, f32=1, groups=4)
            e32.append(max(np.abs(qf[:7] - q1[:7]).max(), (np.abs(vf[:6] - v1[:6]) / np.array([1, 1, 1, 35, 35, 35])).max(), abs(vf[14] - v1[14]) / 180))
            q, v = q1, v1
    print("control steps that start with the mass on a leg: %d; fp64 vs oracle %.2e; fp32 median %.1e max %.1e" % (n_pair, worst, np.median(e32), max(e32)))
    assert n_pair >= 20 and worst < 2e-9 and np.median(e32) < 1e-6 and max(e32) < 1e-5
    # The narrow phase is warm-started from the previous substep's solution (pair_narrow_warm) and falls back, lane by lane, on the cold
    # fixed-count scheme where that does not converge to 64 ulp; the first substep of a control step is always cold.  The comparison above
    # already held both builds against the oracle's cold scheme; here: how often each path ran, in fp32 (what the GPU runs) and in fp64.
    lib.jbh_pair_narrow_stats.argtypes = [C.POINTER(C.c_long), C.c_int]
    st = (C.c_long * 2)()
    lib.jbh_pair_narrow_stats(st, 1)
    P = Ps[bad[0]]
    shares = {}
    for f32 in (1, 0):
        q, v = model.qpos0(P), np.zeros(model.NV)
        rng = np.random.default_rng(3)
        for t in range(12):
            q, v = step(P, q, v, rng.uniform(-1, 1), f32=f32, groups=1)
        lib.jbh_pair_narrow_stats(st, 1)
        shares[f32] = (st[0], st[1])
    print("narrow phase, substeps entirely warm-started / needing the cold scheme: fp32 %s, fp64 %s" % (shares[1], shares[0]))
    assert shares[1][0] > 3 * shares[1][1] and shares[1][1] >= 1          # fp32: mostly warm; at least the control steps' first substeps cold
    assert shares[0][0] + shares[0][1] > 0


def test_contacts_beyond_the_row_cache_give_the_same_answer(params):
    """A build with a row cache of TWO contacts (-DJB_ROW_K=2): nearly every contact then goes the beyond-the-cache way - its candidate
    parked in the overflow store, its rows recomputed in registers in every Newton pass, no rank-one passes on it - in the ordinary and
    the LEAN layout, with one and four lane groups.  Same minimiser as the oracle (and as the default build) to round-off, walking and
    lying on the legs."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build(row_k=2))
    dp = C.POINTER(C.c_double)
    for f in (lib.jbh_step_groups, lib.jbh_step_lean):
        f.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    P = np.ascontiguousarray(params)

    def run(fn, q, v, u, groups):
        q, v, fail = q.copy(), v.copy(), np.zeros(1)
        assert fn(P.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), float(u), 50, 1, 20, 0, groups, 1, fail.ctypes.data_as(dp)) == 0
        assert fail[0] == 0
        return q, v

    worst = 0.0
    for tipped in (False, True):
        env = _contact_states(params, 8, tipped)
        rng = np.random.default_rng(3)
        for t in range(3):
            a = rng.uniform(-1, 1, size=8)
            q0, v0, _ = env.get_state()
            env.step(a, auto_reset=False)
            q1, v1, _ = env.get_state()
            for i in range(8):
                for fn in (lib.jbh_step_groups, lib.jbh_step_lean):
                    for groups in (1, 4):
                        qh, vh = run(fn, q0[i], v0[i], a[i], groups)
                        worst = max(worst, np.abs(qh - q1[i]).max(), (np.abs(vh - v1[i]) / (1 + np.abs(v1[i]))).max())
    # ... and robots lying on the floor (8 contacts or so): with four lane groups the TWO cached leg slots are swept in spread mode, the
    # other six go through the ordinary loop with their rows recomputed per pass - the mixed path of a plan that overflows the cache
    env = _lying_states(params, 8)
    q0, v0, _ = env.get_state()
    a = np.linspace(-1, 1, 8)
    env.step(a, auto_reset=False)
    q1, v1, _ = env.get_state()
    for i in range(8):
        for fn in (lib.jbh_step_groups, lib.jbh_step_lean):
            qh, vh = run(fn, q0[i], v0[i], a[i], 4)
            worst = max(worst, np.abs(qh - q1[i]).max(), (np.abs(vh - v1[i]) / (1 + np.abs(v1[i]))).max())
    assert worst < 1e-10, worst


# ---------------------------------------------------------------------------------------------- the fused rollout's loop on the host
@pytest.fixture(scope="module")
def hrollout():
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.jbh_rollout.argtypes = [dp, dp, dp, dp, ip, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, dp, C.c_int, C.c_int, dp]
    P0 = np.ascontiguousarray(model.default_params(), dtype=np.float64)

    def run(q, v, tgt, counters, K, actions, task, step_limit, seed, env_global, ngroups=1, f32=0, policy_params=None, auto_reset=1):
        D = model.OBS_DIM[task]
        q, v, tgt = q.copy(), v.copy(), tgt.copy()
        cnt = np.array(counters, dtype=np.int32)
        rows = np.zeros((K, D + 2))
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float64)
        pp = None if policy_params is None else np.ascontiguousarray(policy_params, dtype=np.float64)
        rc = lib.jbh_rollout(P0.ctypes.data_as(dp), q.ctypes.data_as(dp), v.ctypes.data_as(dp), tgt.ctypes.data_as(dp), cnt.ctypes.data_as(ip), K,
                             None if a is None else a.ctypes.data_as(dp), model.TASKS.index(task), 50, step_limit, auto_reset, 1, seed, env_global,
                             None if pp is None else pp.ctypes.data_as(dp), ngroups, f32, rows.ctypes.data_as(dp))
        assert rc == 0, rc
        return rows, q, v, tgt, cnt
    return run


@pytest.mark.parametrize("task", ["move_from_origin", "move_to_pose", "move_in_direction"])
def test_rollout_loop_fp64_equals_oracle_through_auto_resets(hrollout, params, task):
    """The K-step loop of the fused rollout kernel (state, step counter, episode and target carried between control steps,
    jb_step.hpp control_step_tail after every 50 substeps) on the host in fp64 against the oracle stepping one control step at a time:
    30 steps of an action tape with a 12-step time limit - two in-loop episode resets (new Philox draws, new target)."""
    seed, env_global, limit, K = 11, 5, 12, 30
    env = O.OracleEnv(1, task, params, step_limit=limit, seed=seed, env_offset=env_global)
    env.reset()
    q, v, t = env.get_state()
    sc, ep = env.counters()
    rng = np.random.default_rng(4)
    tape = rng.uniform(-1, 1, size=K)
    D = model.OBS_DIM[task]
    ref = np.zeros((K, D + 2))
    for k in range(K):
        ob, rw, dn = env.step(tape[k])
        ref[k, :D], ref[k, D], ref[k, D + 1] = ob[0], rw[0], dn[0]
    qo, vo, to = env.get_state()
    sco, epo = env.counters()
    rows, qh, vh, th, cnt = hrollout(q[0], v[0], t[0], [sc[0], ep[0]], K, tape, task, limit, seed, env_global)
    assert ref[:, D + 1].sum() == 2 and np.array_equal(rows[:, D + 1], ref[:, D + 1])
    np.testing.assert_allclose(rows, ref, rtol=0, atol=2e-8)
    np.testing.assert_allclose(qh, qo[0], rtol=0, atol=1e-8)
    np.testing.assert_allclose(th, to[0], rtol=0, atol=1e-12)
    assert cnt[0] == sco[0] and cnt[1] == epo[0]
    # the wave layout of the kernel (main + replica + two helper groups: the replica runs the tail too, and must reset with the main lanes)
    rows4, q4, v4, t4, cnt4 = hrollout(q[0], v[0], t[0], [sc[0], ep[0]], K, tape, task, limit, seed, env_global, ngroups=4)
    np.testing.assert_allclose(rows4, rows, rtol=0, atol=1e-9)
    assert np.array_equal(cnt4, cnt)
    # one K-step call = K one-step calls with the state exported / imported in between (only the contact solver's warm start is lost
    # at each hand-over: the minimiser is the same to the solver's tolerance)
    qs, vs, ts, cs = q[0], v[0], t[0], [sc[0], ep[0]]
    rows1 = np.zeros_like(rows)
    for k in range(K):
        r, qs, vs, ts, cs = hrollout(qs, vs, ts, cs, 1, tape[k:k + 1], task, limit, seed, env_global)
        rows1[k] = r[0]
    np.testing.assert_allclose(rows1, rows, rtol=0, atol=2e-8)


def test_rollout_loop_in_loop_policy_fp64_equals_oracle_with_the_python_policy(hrollout, params):
    """actions = NULL: the heuristic policy evaluated inside the loop on the observation just produced (what the fused kernel does
    instead of a jb_policy_kernel launch per step) against the oracle driven by jitterbug_amd.heuristic_policies.policy_batch."""
    from jitterbug_amd import heuristic_policies as HP
    kw = dict(kick_angle=0.6, speed=0.8, angle_threshold=0.3)
    for task in ("move_from_origin", "move_to_position", "move_to_pose"):
        seed, env_global, limit, K = 2, 9, 15, 25
        env = O.OracleEnv(1, task, params, step_limit=limit, seed=seed, env_offset=env_global)
        ob = env.reset()
        q, v, t = env.get_state()
        sc, ep = env.counters()
        D = model.OBS_DIM[task]
        ref = np.zeros((K, D + 2))
        acts = []
        for k in range(K):
            a = HP.policy_batch(task, ob, **kw)
            acts.append(float(a[0]))
            ob, rw, dn = env.step(a)
            ref[k, :D], ref[k, D], ref[k, D + 1] = ob[0], rw[0], dn[0]
        rows, *_ = hrollout(q[0], v[0], t[0], [sc[0], ep[0]], K, None, task, limit, seed, env_global, policy_params=[kw["kick_angle"], kw["speed"], kw["angle_threshold"]])
        assert len(set(np.sign(acts))) == 2 and ref[:, D + 1].sum() == 1          # the policy really switches, and an episode ends inside
        np.testing.assert_allclose(rows, ref, rtol=0, atol=2e-8)


# ---- the line-searched second solve (jb_sim.hpp newton_phase<LS = true>: MuJoCo's Newton solver, reference jitterbug.xml:18 defaults)
def _ls_stats(reset=True):
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    lib.jbh_ls_stats.argtypes = [C.POINTER(C.c_long), C.c_int]
    a = (C.c_long * 4)()
    lib.jbh_ls_stats(a, 1 if reset else 0)
    return list(a)


@pytest.mark.parametrize("groups", [1, 4])
def test_line_searched_solve_fp64_equals_oracle(gstep, params, groups):
    """max_newton = 1: the plain iteration gets ONE check, so every substep whose active set does not repeat at once goes to the line-searched
    solve - which must reach the minimiser the oracle's Newton (exact line search, fp64) reaches: walking and lying robots, main lanes alone and
    with helper groups (the line search's sweeps are shared out like every other sweep)."""
    worst = 0.0
    resolved = 0
    for tipped in (False, True):
        env = _contact_states(params, 10, tipped)
        rng = np.random.default_rng(11)
        _ls_stats()
        for t in range(5):
            a = np.ones(10) if tipped else rng.uniform(-1, 1, size=10)
            q0, v0, _ = env.get_state()
            env.step(a, auto_reset=False)
            q1, v1, _ = env.get_state()
            for i in range(10):
                qg, vg, cap = gstep(q0[i], v0[i], a[i], groups=groups, maxn=1)
                assert cap == 0
                worst = max(worst, np.abs(qg - q1[i]).max(), (np.abs(vg - v1[i]) / (1 + np.abs(v1[i]))).max())
        st = _ls_stats()
        assert st[0] > 200 and st[1] >= st[0] and st[2] > 0 and st[3] == 0, st      # many substeps were solved twice, some line searches shortened a step, none ended at the cap
        resolved += st[0]
    print("line-searched solve vs oracle (%d groups): %.2e over %d re-solved substeps" % (groups, worst, resolved))
    assert worst < 1e-10, worst


def test_line_searched_solve_settles_rounding_level_cycles_fp32(params):
    """Entry states of substeps (captured on MI355X, fp32 words) in which the plain iteration ran into its cap: most are an edge whose residual is
    zero to within the rounding of the solve, so that the two sets' minimisers - a few 1e-6 |y| apart - send the iteration to each other for
    ever; the first two records do that on the host too, line search or not.  The second solve's growing rounding tolerance ends them."""
    import os
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    lib.jbh_substep_record.argtypes = [dp, fp, C.c_int, C.c_int, C.c_int, dp]
    recs = np.load(os.path.join(os.path.dirname(__file__), "golden", "ls_records.npy"))
    P = np.ascontiguousarray(params, dtype=np.float64)
    _ls_stats()
    for groups in (1, 4):
        for r in recs:
            r = np.ascontiguousarray(r, dtype=np.float32)
            fail = np.zeros(1)
            assert lib.jbh_substep_record(P.ctypes.data_as(dp), r.ctypes.data_as(fp), 12, groups, 0, fail.ctypes.data_as(dp)) == 0
            assert fail[0] == 0
    st = _ls_stats()
    assert st[0] >= 4 and st[3] == 0, st          # the cycling records did go through the second solve, and it settled every one
