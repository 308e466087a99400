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
