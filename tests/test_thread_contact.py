"""The second geom-geom contact randomised models can produce: the motor-axis THREAD (geom 20: a cylinder of 1 mm radius on the motor body,
coaxial with the motor hinge, reference jitterbug.xml:105) against the upper-leg cylinders (jitterbug.xml:52, 65, 79, 92; every jitterbug
geom has contype = conaffinity = 1).  With the reference's sigmas (augmented_jitterbug.py:165-241) the thread sits inside a front upper leg
at rest in ~0.03 % of the draws; robots that do it are DRAWN here: the reference's distribution plus a motor offset that
brings the thread to a front leg's shoulder."""
import ctypes as C

import numpy as np
import pytest

from jitterbug_amd import augmented_jitterbug as aj, model
from oracle import oracle as O


def _touching_models(n, seed=0):
    """draws of the reference's distribution whose motor offset is then moved so that the thread overlaps the upper leg of a front leg by a
    few hundredths of a millimetre to half a millimetre at the rest pose (leg 0 or 1 = XML leg2 / leg3, alternating)"""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        off = aj.draw_offsets(rng, modify_legs=True, modify_mass=True)
        leg = len(out) & 1
        sx = 1.0 if leg == 0 else -1.0
        want = -rng.uniform(2e-5, 5e-4)                    # overlap asked for
        lo, hi = 0.0, 1.0                                  # move the motor axis along the line towards that leg's shoulder end
        base = off[27:29].copy()
        target = np.array([sx * 0.0046, 0.0068])
        ok = False
        for it in range(40):
            mid = 0.5 * (lo + hi)
            off[27:29] = base + mid * (target - base)
            P = model.compile_spec(aj.apply_offsets(off, modify_legs=True, modify_mass=True))
            d = O.pair_thread_geometric(P, model.qpos0(P), leg)[0]
            if d > want:
                lo = mid
            else:
                hi = mid
            if abs(d - want) < 2e-6:
                ok = True
                break
        if ok:
            out.append((P, leg, d))
    return out


@pytest.fixture(scope="module")
def touching():
    return _touching_models(12, seed=7)


def test_thread_flag_keeps_its_sign_apart_from_the_no_ellipsoid_sentinel(touching):
    """the 'this leg can come near the thread' flag is the SIGN of the lane table's LM_PE_IS + 1 entry, whose magnitude is the mass
    pair's broad-phase scale - or, for a mass geom that is not an ellipsoid, a sentinel that used to be -1 and inverted the flag.  Checked on
    the lane tables (host build of the model builder): nominal model - no flag on any leg; a robot drawn to touch - flag on its leg; both
    again with geom 21 turned into a cylinder: the same flags, the mass pair off (x <= 0)."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_lane_table.argtypes = [dp, C.c_int, dp]
    n = lib.jbh_lm_count()

    def pe_is(P, leg):
        out = np.zeros(n)
        assert lib.jbh_lane_table(np.ascontiguousarray(P).ctypes.data_as(dp), leg, out.ctypes.data_as(dp)) == 0
        return out[n - 11:n - 8]          # LM_PE_IS (3): followed by LM_PT_C (3), LM_PT_AX (3), LM_PT_R, LM_PT_H

    P0 = model.default_params()
    Pt, leg, _ = touching[0]
    for P, flagged_leg in ((P0, None), (Pt, leg)):
        flags = {}
        for variant in ("ellipsoid", "cylinder"):
            Q = np.array(P, dtype=np.float64)
            if variant == "cylinder":
                Q[model.P_GEOM + 21 * model.GEOM_STRIDE + model.G_TYPE] = model.GEOM_CYLINDER
            for l in range(3):          # (leg 3's table holds the mass geom itself and refuses a non-ellipsoid: that model is rejected as a whole)
                v = pe_is(Q, l)
                flags[variant, l] = bool(v[1] < 0)
                assert (v[0] > 0) == (variant == "ellipsoid") and abs(v[1]) > 0, (variant, l, v)
        assert all(flags["ellipsoid", l] == flags["cylinder", l] for l in range(3)), flags          # the sentinel does not touch the flag
        if flagged_leg is None:
            assert not any(flags.values()), flags                    # the nominal thread stays 7 mm clear of every leg
        else:
            assert flags["ellipsoid", flagged_leg], flags            # (the flag is conservative: a neighbouring leg may carry it too)


def test_thread_narrow_phase_against_exact_gjk():
    """pair_thread_geometric (the leg as its axis segment, the thread as a flat-capped cylinder in closed form): separated, and with the nearest
    point of the leg's axis inside the segment, the gap IS the exact GJK distance of the two cylinders; at the very end of the leg it is the
    rounded end's (up to the leg's radius - 0.61 mm - short of the flat cap's: the upper legs' ends lie inside the shoulder / knee geometry,
    the convention of the mass pair).  More bisection steps than the kernel's fixed count change nothing above round-off."""
    rng = np.random.RandomState(3)
    inner = end = 0
    for k in range(60):
        off = aj.draw_offsets(rng, modify_legs=True, modify_mass=True)
        off[27:29] += rng.uniform(-1, 1, size=2) * np.array([0.004, 0.005])
        P = model.compile_spec(aj.apply_offsets(off, modify_legs=True, modify_mass=True))
        q = model.qpos0(P)
        q[7:15] = rng.normal(size=8) * 0.05
        q[15] = rng.uniform(0, 2 * np.pi)
        for leg in range(4):
            d, n, pos, segd = O.pair_thread_geometric(P, q, leg)
            d2 = O.pair_thread_geometric(P, q, leg, steps=60)[0]
            assert abs(d - d2) < 1e-9
            g = O.geom_distance(P, q, 4 + 4 * leg, 20)
            if d <= 0:
                assert g == 0.0 or g < 6.2e-4          # overlapping (GJK reports 0), or only the rounded end does
                continue
            assert abs(np.linalg.norm(n) - 1) < 1e-12
            if abs(d - g) < 1e-9:
                inner += 1
            else:
                assert -1e-9 <= g - d <= 6.2e-4, (d, g)          # the capsule end protrudes by at most the leg's radius
                end += 1
    assert inner > 60 and end > 0, (inner, end)


def test_thread_contact_in_the_dynamics_fp64_host_equals_oracle_and_fp32_is_close(touching):
    """The kernel source (PAIR instantiation) on the host against the oracle on robots whose thread rubs a front leg: fp64 to round-off with one
    and with four lane groups, fp32 within the north-star tolerance; and the contact really acts (the oracle without it moves differently)."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_pair.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    worst64, errs32, seen, acts = 0.0, [], 0, 0
    for P, leg, d0 in touching[:6]:
        env = O.OracleEnv(1, "move_from_origin", P, seed=3, random_pose=False)
        env.reset()
        q_no, v_no = None, None
        for t in range(12):
            u = 0.7 if (t // 4) % 2 == 0 else -0.6
            q0, v0, _ = env.get_state()
            dbg = O.forward_debug(P, q0[0], v0[0], u)
            seen += int(any(int(g) >= model.NGEOM + 4 for g in dbg["con_geom"][:dbg["ncon"]]))
            env.step(np.full(1, u), auto_reset=False)
            q1, v1, _ = env.get_state()
            qn, vn = O.step_physics(P, q0[0], v0[0], u, 50, O.default_opts(pair_contacts=1))          # the mass pair only
            acts += int(np.abs(vn - v1[0]).max() > 1e-3)
            for ng in (1, 4):
                qq, vv, fail = q0[0].copy(), v0[0].copy(), np.zeros(1)
                assert lib.jbh_step_pair(P.ctypes.data_as(dp), qq.ctypes.data_as(dp), vv.ctypes.data_as(dp), u, 50, 1, 12, 0, ng, 1, fail.ctypes.data_as(dp)) == 0
                assert fail[0] == 0
                worst64 = max(worst64, np.abs(qq - q1[0]).max(), (np.abs(vv - v1[0]) / (1 + np.abs(v1[0]))).max())
            qq, vv, fail = q0[0].copy(), v0[0].copy(), np.zeros(1)
            assert lib.jbh_step_pair(P.ctypes.data_as(dp), qq.ctypes.data_as(dp), vv.ctypes.data_as(dp), u, 50, 1, 12, 1, 4, 1, fail.ctypes.data_as(dp)) == 0
            errs32.append(max(np.abs(qq[:7] - q1[0][:7]).max(), (np.abs(vv[:6] - v1[0][:6]) / np.array([1, 1, 1, 35, 35, 35])).max()))
    print("thread contact: host fp64 vs oracle %.2e, fp32 median %.2e / max %.2e; substeps starts with the contact %d, env-steps it changes %d" % (worst64, np.median(errs32), np.max(errs32), seen, acts))
    assert seen > 30 and acts > 30
    assert worst64 < 1e-10
    assert np.median(errs32) < 2e-6 and np.quantile(errs32, 0.9) < 2e-5


def test_both_pairs_of_one_leg_at_once_fp64_host_equals_oracle(touching):
    """A robot whose thread reaches a front leg has its motor axis ~7 mm nearer that leg than nominal: when the motor turns, the eccentric mass
    strikes the same leg.  The thread contact (slot 28) and the mass contact (slot 29) are then live in the same substeps - two pair slots,
    two frames, one shoulder - motor cross term that both add to: the kernel source on the host in fp64 against the oracle, one and four
    lane groups."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_pair.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    worst, both = 0.0, 0
    for P, leg, d0 in touching[:4]:
        env = O.OracleEnv(1, "move_from_origin", P, seed=3, random_pose=False)
        env.reset()
        for t in range(25):
            u = 0.8
            q0, v0, _ = env.get_state()
            for sub in (0, 17, 33):          # the contact sets at a few substep offsets of this control step
                qs, vs = O.step_physics(P, q0[0], v0[0], u, sub) if sub else (q0[0], v0[0])
                d = O.forward_debug(P, qs, vs, u)
                g = [int(x) for x in d["con_geom"][:d["ncon"]]]
                both += int(any(model.NGEOM <= x < model.NGEOM + 4 for x in g) and any(x >= model.NGEOM + 4 for x in g))
            env.step(np.full(1, u), auto_reset=False)
            q1, v1, _ = env.get_state()
            for ng in (1, 4):
                qq, vv, fail = q0[0].copy(), v0[0].copy(), np.zeros(1)
                assert lib.jbh_step_pair(P.ctypes.data_as(dp), qq.ctypes.data_as(dp), vv.ctypes.data_as(dp), u, 50, 1, 12, 0, ng, 1, fail.ctypes.data_as(dp)) == 0
                assert fail[0] == 0
                worst = max(worst, np.abs(qq - q1[0]).max(), (np.abs(vv - v1[0]) / (1 + np.abs(v1[0]))).max())
    print("mass + thread contacts of one leg live together in %d sampled substeps; host fp64 vs oracle %.2e" % (both, worst))
    assert both >= 5
    assert worst < 1e-9


def test_captured_double_contact_states_with_lane_groups_fp64_host_equals_oracle():
    """Six env-steps captured from a GPU run (tests/golden/thread_double_contact_states.npy: step, env, action, qpos, qvel, target; models =
    _touching_models(16, seed=11)[env % 16]) in which the mass and the thread touch one leg together.  With helper lane groups the two pair
    slots sit in DIFFERENT groups, and the groups' active-set records are added up: with a plain sum of the 5-bit fields an edge entering one
    contact's set cancelled an edge leaving the other's, the check saw "no change" and the Newton iteration stopped one pass early (errors of
    1e-3 on these states).  The record now carries a per-slot multiplier (jb_sim.hpp contact_apply): 1, 2 and 4 groups = oracle."""
    import os
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    lib.jbh_step_pair.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    models = _touching_models(16, seed=11)
    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "thread_double_contact_states.npy"))
    worst = 0.0
    for row in S:
        i, a = int(row[1]), row[2]
        q, v = row[3:19], row[19:34]
        P = models[i % 16][0]
        q1, v1 = O.step_physics(P, q, v, a, 50)
        for ng in (1, 2, 4):
            qq, vv, fail = q.copy(), v.copy(), np.zeros(1)
            assert lib.jbh_step_pair(P.ctypes.data_as(dp), qq.ctypes.data_as(dp), vv.ctypes.data_as(dp), a, 50, 1, 12, 0, ng, 1, fail.ctypes.data_as(dp)) == 0
            assert fail[0] == 0
            q1n = q1.copy(); q1n[3:7] /= np.linalg.norm(q1n[3:7])
            worst = max(worst, np.abs(qq - q1n).max(), (np.abs(vv - v1) / (1 + np.abs(v1))).max())
    print("captured double-contact env-steps: host fp64 (1 / 2 / 4 lane groups) vs oracle %.2e" % worst)
    assert worst < 1e-9


def test_every_kernel_variant_on_random_orientations_fp64_host_equals_oracle():
    """Robots whose thread rubs a leg (and the nominal robot) thrown on the floor in random orientations - up to a dozen floor contacts of legs,
    box vertices, screws and motor-body geoms next to the geom-geom contacts, four substeps: the kernel source in all four layouts
    (ordinary / LEAN, each with and without the pair contacts; LEAN keeps its overflow candidates and the thread contact's frame outside the
    scratch) with one and four lane groups, fp64 on the host, against the oracle (the layouts without the pair contacts against the oracle
    with them switched off)."""
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    dp = C.POINTER(C.c_double)
    names = ("jbh_step_groups", "jbh_step_lean", "jbh_step_pair", "jbh_step_pair_lean")
    for name in names:
        getattr(lib, name).argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    models = [m[0] for m in _touching_models(6, seed=11)] + [model.default_params()]
    rng = np.random.default_rng(9)
    worst, pair_seen, most = 0.0, 0, 0
    for trial in range(49):
        params = models[trial % len(models)]
        P = np.ascontiguousarray(params, dtype=np.float64)
        q = model.qpos0(params)
        q[3:7] = rng.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
        q[7:15] = rng.normal(size=8) * 0.03; q[15] = rng.uniform(-3, 3)
        v = rng.normal(size=15) * np.array([.05] * 3 + [1] * 3 + [1] * 8 + [20])
        lo, hi = -0.1, 0.2
        for _ in range(30):          # the height at which the first FLOOR contact appears
            mid = 0.5 * (lo + hi); q[2] = mid
            d = O.forward_debug(params, q, v, 0.0)
            if any(int(g) < model.NGEOM for g in d["con_geom"][:d["ncon"]]):
                lo = mid
            else:
                hi = mid
        q[2] = lo - rng.uniform(0.0002, 0.004)
        u = rng.uniform(-1, 1)
        d = O.forward_debug(params, q, v, u)
        pair_seen += any(int(g) >= model.NGEOM for g in d["con_geom"][:d["ncon"]]); most = max(most, d["ncon"])
        with_pairs = O.step_physics(params, q, v, u, 4)
        without = O.step_physics(params, q, v, u, 4, O.default_opts(pair_contacts=0))
        for name in names:
            ref = with_pairs if "pair" in name else without
            for groups in (1, 4):
                qq, vv, fail = q.copy(), v.copy(), np.zeros(1)
                assert getattr(lib, name)(P.ctypes.data_as(dp), qq.ctypes.data_as(dp), vv.ctypes.data_as(dp), float(u), 4, 1, 12, 0, groups, 1, fail.ctypes.data_as(dp)) == 0
                assert fail[0] == 0
                worst = max(worst, np.abs(qq - ref[0]).max(), (np.abs(vv - ref[1]) / (1 + np.abs(ref[1]))).max())
    print("four layouts x 1 / 4 lane groups on random orientations: host fp64 vs oracle %.2e (states with a geom-geom contact %d of 49, most contacts at once %d)" % (worst, pair_seen, most))
    assert pair_seen >= 30 and most >= 8
    assert worst < 1e-10


def _small_actions(rng, n):
    """The motor held within a few degrees of its rest angle: on these robots - their motor axis sits 7 mm nearer a front leg than nominal - a
    turning mass would strike that leg too (by millimetres: the deep-overlap class of tests/test_pair_contact.py); held back, the thread is
    the only geom-geom contact, which is what this file is about."""
    return rng.uniform(-0.02, 0.02, size=n)


@pytest.mark.gpu
def test_gpu_thread_contact_matches_the_oracle():
    """One model per env, every one a robot whose thread rubs a front upper leg: the PAIR kernel (chosen automatically for per-env models) and
    LEAN + PAIR against the oracle, teacher-forced, 150 control steps - the north-star tolerance on every entry of every well-conditioned
    env-step (the thread contact's own activation margin is part of the conditioning); the oracle's contact list shows the thread pair live
    and the mass pair not; and the same robots on the kernel WITHOUT the pair contacts (JB_FLAG_NO_PAIR) leave the tolerance."""
    from jitterbug_amd.vec_env import JitterbugVecEnv
    models = _touching_models(16, seed=11)
    n = 64
    P = np.stack([models[i % len(models)][0] for i in range(n)])
    for flags, name in ((0, "pair"), (2, "lean_pair"), (8, "ordinary")):
        g = JitterbugVecEnv(n, "move_to_pose", seed=4, auto_reset=False, params=P, flags=flags, envs_per_wave=4 if flags == 2 else 0)
        assert g.kernel_variant == name
        o = O.OracleEnv(n, "move_to_pose", P, seed=4, per_env_model=True)
        g.reset(); o.reset()
        rng = np.random.default_rng(4)
        well_bad = well_tot = off = 0
        worst = 0.0
        thread_live = mass_live = 0
        for t in range(150):
            a = _small_actions(rng, n)
            q, v, tg = o.get_state()
            if t % 25 == 0:
                for i in range(0, n, 8):
                    d = O.forward_debug(P[i], q[i], v[i], a[i])
                    geoms = [int(x) for x in d["con_geom"][:d["ncon"]]]
                    thread_live += any(x >= model.NGEOM + 4 for x in geoms)
                    mass_live += any(model.NGEOM <= x < model.NGEOM + 4 for x in geoms)
            g.set_state(q, v, tg)
            og, rg, dg, _ = g.step(a)
            oo, ro, do = o.step(a, auto_reset=False)
            well = o.margins() >= 3e-8
            err = np.abs(og.astype(np.float64) - oo)
            w = err <= 1e-4 * np.abs(oo) + 1e-6
            well_bad += int((~w[well]).sum()); well_tot += int(w[well].size)
            off += int((err > 1e-3 * np.abs(oo) + 1e-5).any(axis=1).sum())
            if well.any():
                worst = max(worst, float(err[well].max()))
        sc, ep, cap = g.counters()
        g.close()
        print("thread-touching models, %s kernel: %d of %d well-conditioned entries outside the tolerance (worst %.1e), env-steps far off %d; oracle samples with the thread pair live %d, with the mass pair live %d"
              % (name, well_bad, well_tot, worst, off, thread_live, mass_live))
        assert thread_live >= 40 and mass_live == 0
        assert cap.sum() == 0
        if flags == 8:
            assert off > 2000, off                      # without the contact the thread passes through the leg
        else:
            assert well_bad <= 3 and worst < 2e-5 and well_tot > 0.5 * n * 150 * 19, (well_bad, worst, well_tot)


@pytest.mark.gpu
def test_gpu_thread_and_mass_contacts_together():
    """The same robots with the motor turning: the mass strikes the leg the thread already rubs - both pair slots of one lane live in the
    same substeps (tests/test_thread_contact.py::test_both_pairs_... holds the kernel source to the oracle in fp64 there).  On the GPU 1 % of
    these env-steps are ill-conditioned by the oracle's own margin (the leg's axis inside the mass); the well-conditioned entries hold the
    north-star tolerance but for a handful (below 2e-5), nothing diverges, every solve converges - PAIR and LEAN + PAIR."""
    from tests.test_gpu_parity import _teacher_forced
    models = _touching_models(16, seed=11)
    P = np.stack([models[i % len(models)][0] for i in range(64)])
    for flags in (0, 2):
        r = _teacher_forced("move_to_pose", 64, 150, seed=4, params=P, flags=flags)
        print("thread-touching models, motor turning, flags %d:" % flags, r)
        assert r["well_bad"] <= 6 and r["worst_well"] < 2e-5 and r["well_big"] == 0 and r["cap"] == 0, r      # measured: 2 (PAIR) / 1 (LEAN + PAIR), worst 6.5e-6
        assert r["frac"] >= 0.999, r
