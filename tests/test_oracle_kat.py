"""Known-answer tests pinning the CPU oracle.

The reference ships no golden vectors and its physics engine (MuJoCo 2.0 via
dm_control) is not available, so these are the closed-form pins derivable from
the reference's own text, plus internal consistency checks of the restatement
(conservation laws converge with the timestep, two independent contact solvers
agree, KKT conditions hold)."""
import math

import numpy as np
import pytest

from jitterbug_amd import model
from oracle import oracle as O


def _random_state(rng, P, scale=1.0):
    q = model.qpos0(P)
    q[3:7] = rng.normal(size=4)
    q[3:7] /= np.linalg.norm(q[3:7])
    q[7:15] = rng.normal(size=8) * 0.05 * scale
    q[15] = rng.uniform(-3, 3)
    v = rng.normal(size=15) * np.array([.1] * 3 + [2] * 3 + [3] * 8 + [50]) * scale
    return q, v


def test_free_fall(params):
    # contacts off, u=0, from rest, 50 substeps: vz = -g*50h ; z = z0 - g h^2 * 50*51/2 ; hinges stay 0
    o = O.default_opts(contacts=0)
    q, v = O.step_physics(params, model.qpos0(params), np.zeros(15), 0.0, 50, o)
    assert v[2] == pytest.approx(-0.0981, abs=1e-12)
    assert q[2] == pytest.approx(0.035 - 9.81 * 0.0002 ** 2 * (50 * 51 / 2), abs=1e-12)
    assert np.abs(q[7:]).max() < 1e-12 and np.abs(v[3:]).max() < 1e-10
    np.testing.assert_allclose(q[3:7], [1, 0, 0, 0], atol=1e-12)


def test_mass_matrix_matches_numpy_at_qpos0(params):
    d = O.forward_debug(params, model.qpos0(params), np.zeros(15), 0.0, O.default_opts(contacts=0))
    M, _, _ = model.mass_matrix_qpos0(params)
    np.testing.assert_allclose(d["M"], M, rtol=0, atol=1e-18)
    assert np.abs(d["bias"][2] - 1.75165e-2 * 9.81) < 1e-6       # gravity on the z dof
    # M is frame invariant in its translational block and symmetric positive definite anywhere
    q, v = _random_state(np.random.default_rng(3), params)
    d = O.forward_debug(params, q, v, 0.3, O.default_opts(contacts=0))
    np.testing.assert_allclose(d["M"][:3, :3], np.eye(3) * d["M"][0, 0], atol=1e-15)
    np.testing.assert_allclose(d["M"], d["M"].T, atol=0)
    assert np.linalg.eigvalsh(d["M"]).min() > 0


def test_kinetic_energy_equals_half_qMq(params):
    q, v = _random_state(np.random.default_rng(5), params)
    d = O.forward_debug(params, q, v, 0.0, O.default_opts(contacts=0))
    me = O.momentum_energy(params, q, v)
    assert me["T"] == pytest.approx(0.5 * v @ d["M"] @ v, rel=1e-12)


def test_momentum_drift_is_first_order_in_h(params):
    # free flight, gravity + contacts off, motor driven: total momentum is conserved by the
    # continuous system; semi-implicit Euler keeps it to O(h) -> halving h halves the drift.
    P = params.copy()
    P[model.P_GRAVITY:model.P_GRAVITY + 3] = 0
    q, v = _random_state(np.random.default_rng(0), P)
    o = O.default_opts(contacts=0)
    drift = []
    for h in (2e-4, 1e-4, 5e-5):
        P[model.P_TIMESTEP] = h
        m0 = O.momentum_energy(P, q, v)
        q1, v1 = O.step_physics(P, q, v, 0.8, int(round(0.02 / h)), o)
        m1 = O.momentum_energy(P, q1, v1)
        drift.append((np.abs(m1["P"] - m0["P"]).max(), np.abs(m1["L"] - m0["L"]).max()))
    for k in range(2):
        assert drift[0][k] / drift[1][k] == pytest.approx(2.0, rel=0.1)
        assert drift[1][k] / drift[2][k] == pytest.approx(2.0, rel=0.1)
    assert drift[0][0] / np.abs(O.momentum_energy(P, q, v)["P"]).max() < 0.06


def test_energy_drift_is_first_order_in_h(params):
    # conservative system (no damping, no actuator, gravity + springs on): dE -> 0 linearly with h.
    P = params.copy()
    P[model.P_GEAR] = 0
    for h in range(9):
        P[model.P_HINGE + h * model.HINGE_STRIDE + model.H_DAMPING] = 0
    q, v = _random_state(np.random.default_rng(0), P)
    o = O.default_opts(contacts=0)
    dE = []
    for h in (1e-4, 5e-5, 2.5e-5):
        P[model.P_TIMESTEP] = h
        e0 = O.momentum_energy(P, q, v)
        q1, v1 = O.step_physics(P, q, v, 0.0, int(round(0.02 / h)), o)
        e1 = O.momentum_energy(P, q1, v1)
        dE.append(abs(e1["T"] + e1["V"] - e0["T"] - e0["V"]))
    assert dE[0] / dE[1] == pytest.approx(2.0, rel=0.25)
    assert dE[1] / dE[2] == pytest.approx(2.0, rel=0.25)
    assert dE[2] < 1e-5          # vs a total mechanical energy of ~8e-3 J


def test_actuator_free_run_speed(params):
    # tau = 0.00833 u - 5.5511e-5 qd_motor  ->  joint speed settles at 150.06 rad/s for u=1 (clip beyond 1)
    P = params.copy()
    P[model.P_GRAVITY:model.P_GRAVITY + 3] = 0
    o = O.default_opts(contacts=0)
    P0 = P.copy()
    P0[model.P_GEAR] = 0                          # same model without the actuator
    vfree = np.concatenate([np.zeros(14), [0.00833 / (0.8 * 0.00833 ** 2)]])
    assert vfree[14] == pytest.approx(150.06, rel=1e-4)
    with_act = O.forward_debug(P, model.qpos0(P), vfree, 5.0, o)["tau"][14]      # u=5 is clipped to +1
    without = O.forward_debug(P0, model.qpos0(P), vfree, 0.0, o)["tau"][14]
    assert abs(with_act - without) < 1e-15        # zero actuator torque at the free-run speed
    q, v = O.step_physics(P, model.qpos0(P), np.zeros(15), 1.0, 1500, o)
    assert 140 < v[14] < 160                       # spins up to about that speed (body wobble modulates it)
    d = O.forward_debug(P, model.qpos0(P), np.concatenate([np.zeros(14), [10.0]]), 0.5, o)
    d0 = O.forward_debug(P, model.qpos0(P), np.concatenate([np.zeros(14), [10.0]]), 0.0, o)
    assert d["tau"][14] - d0["tau"][14] == pytest.approx(0.00833 * 0.5, rel=1e-12)


def test_obs_and_reward_at_reset_pose(params):
    q, v, t = O.reset(params, "move_from_origin", False, 0, 0, 0)
    np.testing.assert_allclose(q, model.qpos0(params))
    obs = O.observation(params, "move_from_origin", q, v, t)
    np.testing.assert_allclose(obs, [0, 0, -0.3, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.5, 0], atol=1e-15)
    assert O.reward(params, "move_from_origin", q, v, t) == 0.0          # d=0 -> P=1 -> (1-P) U = 0
    assert O.reward(params, "move_to_position", q, v, t) == 1.0


def test_obs_dims(params):
    q, v, t = O.reset(params, "move_to_pose", True, 1, 2, 3)
    for task, D in model.OBS_DIM.items():
        assert O.observation(params, task, q, v, t).shape == (D,)


def test_reward_closed_forms(params):
    # P(d=0.05)=0.1; U(Rzz=0.5)=0.1; H(dpsi=pi/4)=0.5; V(v=0.05)=0.5; V(v>=0.1)=1; H(|dpsi|>=pi/2)=0
    assert O.tolerance(0.05, (0, 0), 0.05) == pytest.approx(0.1, rel=1e-12)
    assert O.tolerance(0.5, (1, 1), 0.5) == pytest.approx(0.1, rel=1e-12)
    assert O.tolerance(math.pi / 4, (0, 0), math.pi / 2, 0.0, "cosine") == pytest.approx(0.5, rel=1e-12)
    assert O.tolerance(-math.pi / 4, (0, 0), math.pi / 2, 0.0, "cosine") == pytest.approx(0.5, rel=1e-12)
    assert O.tolerance(math.pi / 2, (0, 0), math.pi / 2, 0.0, "cosine") == 0.0
    assert O.tolerance(2.0, (0, 0), math.pi / 2, 0.0, "cosine") == 0.0
    assert O.tolerance(0.05, (0.1, math.inf), 0.1, 0.0, "linear") == pytest.approx(0.5, rel=1e-12)
    assert O.tolerance(0.1, (0.1, math.inf), 0.1, 0.0, "linear") == 1.0
    assert O.tolerance(3.0, (0.1, math.inf), 0.1, 0.0, "linear") == 1.0
    assert O.tolerance(-0.2, (0.1, math.inf), 0.1, 0.0, "linear") == 0.0
    # composed through the state: jitterbug 5 cm from the target, upright
    q = model.qpos0(params)
    q[0] = 0.05
    terms = O.reward_terms(params, q, np.zeros(15), np.zeros(3))
    assert terms["P"] == pytest.approx(0.1, rel=1e-12) and terms["U"] == 1.0
    assert O.reward(params, "move_from_origin", q, np.zeros(15), np.zeros(3)) == pytest.approx(0.9, rel=1e-12)
    # tilt so that Rzz = 0.5 (60 deg about x)
    q = model.qpos0(params)
    q[3:7] = [math.cos(math.pi / 6), math.sin(math.pi / 6), 0, 0]
    assert O.reward_terms(params, q, np.zeros(15), np.zeros(3))["U"] == pytest.approx(0.1, rel=1e-9)
    # heading: jitterbug yaw = atan2(R10,R00) - pi/2; target psi = -pi/2 + pi/4 -> dpsi = pi/4
    q = model.qpos0(params)
    t = np.array([0, 0, -math.pi / 2 + math.pi / 4])
    assert O.reward_terms(params, q, np.zeros(15), t)["H"] == pytest.approx(0.5, rel=1e-12)
    obs = O.observation(params, "face_direction", q, np.zeros(15), t)
    assert obs[15] == pytest.approx(0.25, rel=1e-12)
    # velocity in target frame: target yaw 90deg, world vy = 0.05 -> v_x^target = 0.05
    v = np.zeros(15)
    v[1] = 0.05
    t = np.array([0, 0, math.pi / 2])
    assert O.reward_terms(params, q, v, t)["V"] == pytest.approx(0.5, rel=1e-9)
    obs = O.observation(params, "move_in_direction", q, v, t)
    np.testing.assert_allclose(obs[16:19], [0.05, 0, 0], atol=1e-15)


def test_wrap_convention(params):
    # angle in (-pi, pi]: motor obs = wrap(q_m + pi/2)/pi
    q = model.qpos0(params)
    q[15] = math.pi / 2                     # -> exactly +pi -> stays +pi -> obs 1.0
    assert O.observation(params, "move_from_origin", q, np.zeros(15), np.zeros(3))[13] == pytest.approx(1.0, abs=1e-15)
    q[15] = math.pi / 2 + 1e-9              # just above pi wraps to just above -pi
    assert O.observation(params, "move_from_origin", q, np.zeros(15), np.zeros(3))[13] == pytest.approx(-1.0, abs=1e-8)
    q[15] = 1234.5
    a = O.observation(params, "move_from_origin", q, np.zeros(15), np.zeros(3))[13] * math.pi
    assert -math.pi < a <= math.pi
    assert math.cos(a) == pytest.approx(math.cos(1234.5 + math.pi / 2), abs=1e-9)


def test_reset_statistics(params):
    n = 4000
    qs, ts = [], []
    for e in range(n):
        q, v, t = O.reset(params, "move_to_pose", True, 7, e, 0)
        qs.append(q), ts.append(t)
        assert np.all(v == 0)
    qs, ts = np.array(qs), np.array(ts)
    rad = np.hypot(ts[:, 0], ts[:, 1])
    assert rad.min() >= 0.05 and rad.max() < 0.2 and abs(rad.mean() - 0.125) < 0.005
    assert ts[:, 2].min() >= 0 and ts[:, 2].max() < 2 * math.pi and abs(ts[:, 2].mean() - math.pi) < 0.1
    th = 2 * np.arctan2(np.linalg.norm(qs[:, 4:7], axis=1), qs[:, 3])
    assert th.min() >= 0 and th.max() <= 2 * math.pi and abs(th.mean() - math.pi) < 0.1
    axis = qs[:, 4:7] / np.linalg.norm(qs[:, 4:7], axis=1, keepdims=True)
    assert np.abs(axis[:, 0] / axis[:, 2]).max() <= 0.025 and np.abs(axis[:, 1] / axis[:, 2]).max() <= 0.025
    np.testing.assert_allclose(np.linalg.norm(qs[:, 3:7], axis=1), 1, atol=1e-15)
    np.testing.assert_allclose(qs[:, :3], np.tile([0, 0, 0.035], (n, 1)))
    # task switch: targets only where the reference sets them (jitterbug.py:613-651)
    for task, has_xy, has_psi in (("move_from_origin", 0, 0), ("face_direction", 0, 1), ("move_in_direction", 0, 1),
                                  ("move_to_position", 1, 0), ("move_to_pose", 1, 1)):
        _, _, t = O.reset(params, task, True, 7, 11, 2)
        assert (abs(t[0]) + abs(t[1]) > 0) == bool(has_xy) and (t[2] != 0) == bool(has_psi)
    # streams: same key -> same draw; different env / episode -> different
    a = O.reset(params, "move_to_pose", True, 7, 5, 1)
    b = O.reset(params, "move_to_pose", True, 7, 5, 1)
    c = O.reset(params, "move_to_pose", True, 7, 5, 2)
    assert np.all(a[0] == b[0]) and np.all(a[2] == b[2]) and not np.all(a[0] == c[0])


def test_philox_known_answer():
    # Random123 Philox4x32-10 KAT: counter = key = 0 ; and the all-ones vector
    np.testing.assert_array_equal(O.philox(0, 0, 0, 0), np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], dtype=np.uint32))
    np.testing.assert_array_equal(O.philox(0xffffffffffffffff, 0xffffffffffffffff, 0xffffffff, 0xffffffff),
                                  np.array([0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd], dtype=np.uint32))


def test_standing_equilibrium_and_kkt(params):
    q, v = model.qpos0(params), np.zeros(15)
    o = O.default_opts()
    warm = np.zeros(O.WARM_SIZE)
    for _ in range(40):
        q, v = O.step_physics(params, q, v, 0.0, 50, o, warm)
    d = O.forward_debug(params, q, v, 0.0, o)
    assert d["ncon"] == 4 and sorted(d["con_geom"]) == [7, 11, 15, 19]          # the four feet
    assert np.all(d["con_dist"] < 0) and np.all(d["con_dist"] > -0.002)
    # pyramidal edges: net normal force = sum of edge forces ~ weight
    assert d["f"].sum() == pytest.approx(1.75165e-2 * 9.81, rel=0.01)
    assert np.abs(v[:14]).max() < 0.02 and abs(q[2] - 0.033) < 5e-4   # (the free motor hinge still swings)
    # KKT of the dual:  f >= 0,  (J qacc - aref) + R f >= 0,  complementarity
    s = d["jar"] + d["R"] * d["f"]
    assert d["f"].min() >= 0 and s.min() > -1e-9 and np.abs(s * d["f"]).max() < 1e-10
    # regulariser: R = 2 mu^2 (1-d)/d * tran (1+mu^2) with the foot body's invweight0
    tran = params[model.P_BODY + 2 * model.BODY_STRIDE + model.B_INVW_TRAN]
    imp = O.forward_debug(params, q, v, 0.0, o)
    assert d["R"][list(d["con_geom"]).index(7) * 4] < 4 * tran * (1 - 0.9) / 0.9 * 1.0001
    assert d["R"][list(d["con_geom"]).index(7) * 4] > 4 * tran * (1 - 0.95) / 0.95 * 0.9999


def test_newton_and_pgs_agree(params):
    rng = np.random.default_rng(0)
    N = 16
    pgs = O.OracleEnv(N, "move_to_pose", params, seed=3, opts=O.default_opts(solver=0, solver_tol=1e-13))
    newt = O.OracleEnv(N, "move_to_pose", params, seed=3, opts=O.default_opts(solver=1, solver_tol=1e-13))
    pgs.reset(), newt.reset()
    worst = 0
    for t in range(40):
        a = rng.uniform(-1, 1, size=N)
        pgs.set_state(*newt.get_state())
        o1, r1, _ = newt.step(a, auto_reset=False)
        o2, r2, _ = pgs.step(a, auto_reset=False)
        worst = max(worst, np.abs(o1 - o2).max(), np.abs(r1 - r2).max())
    assert worst < 1e-10
    assert newt.stats().ncon_max >= 2


def test_episode_counters_and_auto_reset(params):
    env = O.OracleEnv(3, "move_from_origin", params, seed=1, step_limit=4, nsub=2)
    ob0 = env.reset()
    for t in range(3):
        ob, r, d = env.step(np.zeros(3))
        assert not d.any()
    ob, r, d = env.step(np.zeros(3))          # 4th step: done, VecEnv semantics -> obs of the new episode
    assert d.all()
    sc, ep = env.counters()
    assert np.all(sc == 0) and np.all(ep == 3)      # create = reset #0, reset() = #1, auto-reset = #2
    q, v, t = env.get_state()
    for i in range(3):
        qe, ve, te = O.reset(params, "move_from_origin", True, 1, i, 2)
        np.testing.assert_array_equal(q[i], qe)
        np.testing.assert_allclose(ob[i], O.observation(params, "move_from_origin", qe, ve, te))


# ----------------------------------------------------------------------------------------------- reference-held tables
def _norm_tables():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "norm_tables.json")))


def _raw_observation(P, task, q, v, t):
    """The UN-normalised observation entries, written here independently of the oracle from the reference's accessors
    (jitterbug.py:180-317); the golden tables then normalise them exactly as Jitterbug._norm does (:668-671)."""
    w, x, y, z = q[3:7]
    R = np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
    wrap = lambda a: a - 2 * np.pi * np.ceil((a - np.pi) / (2 * np.pi))          # (-pi, pi]
    raw = list(q[0:7]) + list(v[0:6]) + [wrap(q[15] + np.pi / 2), v[14]]
    psi = t[2]
    Rt = np.array([[np.cos(psi), -np.sin(psi), 0], [np.sin(psi), np.cos(psi), 0], [0, 0, 1]])
    ang = wrap(np.arctan2(Rt[1, 0], Rt[0, 0]) - (np.arctan2(R[1, 0], R[0, 0]) - np.pi / 2))
    ipos = P[model.P_BODY + model.B_COM:model.P_BODY + model.B_COM + 3]
    sensor = v[0:3] + R @ np.cross(v[3:6], ipos)                                  # framelinvel of a "body" object: at xipos
    rel = R.T @ (np.array([t[0], t[1], P[model.P_TARGETZ]]) - q[0:3])
    extra = {"move_from_origin": [], "face_direction": [ang], "move_in_direction": [ang] + list(Rt.T @ sensor),
             "move_to_position": list(rel), "move_to_pose": list(rel) + [ang]}[task]
    return np.array(raw + extra)


@pytest.mark.parametrize("task", model.TASKS)
def test_observation_normalisation_matches_reference_tables(params, task):
    """_NORM_ALL / _NORM_TASKS are literals of the reference (jitterbug.py:324-372), extracted by tools/gen_golden_norm.py:
    the oracle's observation must be those tables applied to the raw accessor values."""
    g = _norm_tables()
    tab = np.array(g["_NORM_ALL"] + g["_NORM_TASKS"][task])
    assert tab.shape == (model.OBS_DIM[task], 2)
    rng = np.random.default_rng(model.TASKS.index(task))
    for _ in range(50):
        q, v = _random_state(rng, params)
        q[:3] += rng.normal(size=3) * [0.5, 0.5, 0.01]
        t = np.array([rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(0, 2 * np.pi)])
        raw = _raw_observation(params, task, q, v, t)
        exp = (raw - tab[:, 0]) / (tab[:, 1] - tab[:, 0]) * 2.0 - 1.0
        np.testing.assert_allclose(O.observation(params, task, q, v, t), exp, rtol=0, atol=1e-12)


def test_reference_task_constants():
    g = _norm_tables()
    from jitterbug_amd import jitterbug as J
    from jitterbug_amd import vec_env
    assert (g["DEFAULT_TIME_LIMIT"], g["DEFAULT_CONTROL_TIMESTEP"], g["TARGET_SPEED"]) == (vec_env.DEFAULT_TIME_LIMIT, vec_env.DEFAULT_CONTROL_TIMESTEP, J.TARGET_SPEED)
    np.testing.assert_array_equal(J.Jitterbug._NORM_ALL, np.array(g["_NORM_ALL"]))
    for k, v in g["_NORM_TASKS"].items():
        np.testing.assert_array_equal(np.asarray(J.Jitterbug._NORM_TASKS[k]).reshape(-1, 2), np.array(v).reshape(-1, 2))
    # velocity reward saturates at TARGET_SPEED (reference :854-866)
    assert O.tolerance(g["TARGET_SPEED"], bounds=(g["TARGET_SPEED"], float("inf")), margin=g["TARGET_SPEED"], value_at_margin=0.0, sigmoid="linear") == 1.0


def test_framelinvel_is_measured_at_the_root_bodys_centre_of_mass(params):
    """The reference's sensor is <framelinvel objtype="body" objname="jitterbug"> (jitterbug.xml:121): MuJoCo measures an
    mjOBJ_BODY object at the body's inertial frame.  Closed form from the XML's four core geoms (masses of SURVEY.md 8a row 1):
    ipos = sum m c / sum m = (0, 6.008 mm, -10.62 mm) from the root origin.  Spinning at 10 rad/s about the body z axis with
    the joint origin at rest, the sensor reads w x ipos = (-0.06008, 0, 0) m/s - comparable to TARGET_SPEED."""
    m = np.array([1.1180e-4, 2.3654e-3, 1.1611e-3, 1.3547e-3])
    c = np.array([[0, 0, .04], [0, 0, .021], [0, .006, .026], [0, .017, .026]]) - [0, 0, .035]
    ipos = (m[:, None] * c).sum(0) / m.sum()
    np.testing.assert_allclose(params[model.P_BODY + model.B_COM:model.P_BODY + model.B_COM + 3], ipos, atol=2e-7)
    q, v, t = model.qpos0(params), np.zeros(15), np.zeros(3)
    v[5] = 10.0
    obs = O.observation(params, "move_in_direction", q, v, t)
    np.testing.assert_allclose(obs[16:19], [-10 * ipos[1], 10 * ipos[0], 0], atol=5e-6)      # hand masses carry 5 digits
    assert obs[16] == pytest.approx(-0.0601, abs=2e-4)
    # rotated body + rotated target frame: v + R (w x ipos), then R_t^T
    rng = np.random.default_rng(1)
    q, v = _random_state(rng, params)
    t = np.array([0.1, -0.05, 1.1])
    raw = _raw_observation(params, "move_in_direction", q, v, t)
    np.testing.assert_allclose(O.observation(params, "move_in_direction", q, v, t)[16:19], raw[16:19], atol=1e-12)
    assert np.abs(raw[16:19] - np.array([[np.cos(1.1), np.sin(1.1), 0], [-np.sin(1.1), np.cos(1.1), 0], [0, 0, 1]]) @ v[:3]).max() > 1e-3
    # the velocity reward sees it too
    q, v = model.qpos0(params), np.zeros(15)
    v[0], v[5] = 0.1, 10.0                              # joint origin at TARGET_SPEED, but the sensor point moves slower
    assert O.reward_terms(params, q, v, np.zeros(3))["V"] == pytest.approx((0.1 - 10 * ipos[1]) / 0.1, abs=1e-4)
