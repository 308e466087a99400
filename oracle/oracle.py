"""ctypes wrapper around the CPU fp64 oracle (oracle/jb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under jitterbug_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libjb_oracle.so")

NQ, NV, NGEOM = 16, 15, 22
NPARAM = 612
WARM_SIZE = NGEOM * 16 + NV + 32          # (jb_oracle.c WARM_SIZE: floor contacts, last qacc, the two geom-geom pairs of each leg)
MAXCON = 64
MAXROW = 4 * MAXCON
TASKS = ("move_from_origin", "face_direction", "move_in_direction", "move_to_position", "move_to_pose")
OBS_DIM = (15, 16, 19, 18, 19)


def usable_cores():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota.  (A GPU box shows all of the
    host's CPUs but hands out a share: an OpenMP team sized by the CPU count would spin on a fraction of that many cores.)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("jb_oracle.c", "jb_clearance.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


class Opts(C.Structure):
    _fields_ = [("contacts", C.c_int), ("implicit_damp", C.c_int), ("solver_iters", C.c_int),
                ("solver_tol", C.c_double), ("warmstart", C.c_int), ("feet_only", C.c_int), ("solver", C.c_int), ("pair_contacts", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("ncon_last", C.c_int), ("ncon_max", C.c_int), ("sweeps_total", C.c_int), ("sweeps_max", C.c_int),
                ("nsolve", C.c_int), ("overflow", C.c_int), ("resid_max", C.c_double), ("margin_min", C.c_double)]


class Debug(C.Structure):
    _fields_ = [("ncon", C.c_int), ("nrow", C.c_int),
                ("con_dist", C.c_double * MAXCON), ("con_pos", (C.c_double * 3) * MAXCON), ("con_geom", C.c_int * MAXCON),
                ("f", C.c_double * MAXROW), ("aref", C.c_double * MAXROW), ("Rdiag", C.c_double * MAXROW), ("jar", C.c_double * MAXROW),
                ("M", C.c_double * (NV * NV)), ("bias", C.c_double * NV), ("tau", C.c_double * NV),
                ("qacc_smooth", C.c_double * NV), ("qacc", C.c_double * NV), ("qfrc_constraint", C.c_double * NV)]


_lib = None
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.jbo_default_opts.argtypes = [C.POINTER(Opts)]
        L.jbo_step_physics.argtypes = [_dp, _dp, _dp, C.c_double, C.c_int, C.POINTER(Opts), _dp, C.POINTER(Stats)]
        L.jbo_forward_debug.argtypes = [_dp, _dp, _dp, C.c_double, C.POINTER(Opts), C.POINTER(Debug)]
        L.jbo_momentum_energy.argtypes = [_dp, _dp, _dp, _dp]
        L.jbo_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.jbo_reset.argtypes = [_dp, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _dp, _dp, _dp]
        L.jbo_observation.argtypes = [_dp, C.c_int, _dp, _dp, _dp, _dp]
        L.jbo_reward.argtypes = [_dp, C.c_int, _dp, _dp, _dp]
        L.jbo_reward.restype = C.c_double
        L.jbo_reward_terms.argtypes = [_dp, _dp, _dp, _dp, _dp]
        L.jbo_tolerance.argtypes = [C.c_double] * 5 + [C.c_int]
        L.jbo_tolerance.restype = C.c_double
        L.jbo_env_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _dp, C.c_int, C.POINTER(Opts)]
        L.jbo_env_create.restype = C.c_void_p
        L.jbo_env_destroy.argtypes = [C.c_void_p]
        L.jbo_env_reset.argtypes = [C.c_void_p, C.c_void_p, _dp]
        L.jbo_env_step.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_void_p, C.c_int, C.c_int]
        L.jbo_env_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.jbo_env_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.jbo_env_get_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.jbo_env_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.jbo_env_get_margin.argtypes = [C.c_void_p, _dp]
        L.jbo_pair_clearance.argtypes = [_dp, _dp, C.c_void_p]
        L.jbo_pair_clearance.restype = C.c_double
        L.jbo_geom_distance.argtypes = [_dp, _dp, C.c_int, C.c_int]
        L.jbo_geom_distance.restype = C.c_double
        L.jbo_num_tested_pairs.argtypes = [_dp]
        L.jbo_pair_clearance_batch.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_void_p, C.c_int]
        L.jbo_mass_sweep_clearance_batch.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _dp]
        L.jbo_geom_world.argtypes = [_dp, _dp, C.c_int, _dp, _dp, _dp]
        assert L.jbo_debug_size() == C.sizeof(Debug), (L.jbo_debug_size(), C.sizeof(Debug))
        L.jbo_set_threads.argtypes = [C.c_int]
        L.jbo_set_threads(usable_cores())
        _lib = L
    return _lib


def default_opts(**kw):
    o = Opts()
    lib().jbo_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def task_id(task):
    return TASKS.index(task) if isinstance(task, str) else int(task)


def step_physics(P, qpos, qvel, ctrl, nsub=50, opts=None, warm=None, stats=None):
    """Advance ONE env in place by nsub substeps. Returns (qpos, qvel) (new arrays)."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.array(qpos, dtype=np.float64).copy()
    v = np.array(qvel, dtype=np.float64).copy()
    opts = opts or default_opts()
    lib().jbo_step_physics(_p(P), _p(q), _p(v), float(ctrl), int(nsub), C.byref(opts),
                           _p(warm) if warm is not None else None, C.byref(stats) if stats is not None else None)
    return q, v


def forward_debug(P, qpos, qvel, ctrl=0.0, opts=None):
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.ascontiguousarray(qpos, dtype=np.float64)
    v = np.ascontiguousarray(qvel, dtype=np.float64)
    d = Debug()
    lib().jbo_forward_debug(_p(P), _p(q), _p(v), float(ctrl), C.byref(opts or default_opts()), C.byref(d))
    out = dict(ncon=d.ncon, nrow=d.nrow,
               M=np.array(d.M).reshape(NV, NV), bias=np.array(d.bias), tau=np.array(d.tau),
               qacc_smooth=np.array(d.qacc_smooth), qacc=np.array(d.qacc), qfrc_constraint=np.array(d.qfrc_constraint),
               con_dist=np.array(d.con_dist[:d.ncon]), con_geom=np.array(d.con_geom[:d.ncon]),
               con_pos=np.array([list(d.con_pos[i]) for i in range(d.ncon)]).reshape(d.ncon, 3),
               f=np.array(d.f[:d.nrow]), aref=np.array(d.aref[:d.nrow]), R=np.array(d.Rdiag[:d.nrow]), jar=np.array(d.jar[:d.nrow]))
    return out


def momentum_energy(P, qpos, qvel):
    out = np.zeros(8)
    lib().jbo_momentum_energy(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)),
                              _p(np.ascontiguousarray(qvel, dtype=np.float64)), _p(out))
    return dict(P=out[0:3], L=out[3:6], T=out[6], V=out[7])


def philox(seed, env, episode, stream):
    out = (C.c_uint32 * 4)()
    lib().jbo_philox(seed, env, episode, stream, out)
    return np.array(list(out), dtype=np.uint32)


def reset(P, task, random_pose, seed, env, episode):
    P = np.ascontiguousarray(P, dtype=np.float64)
    q, v, t = np.zeros(NQ), np.zeros(NV), np.zeros(3)
    lib().jbo_reset(_p(P), task_id(task), int(random_pose), seed, env, episode, _p(q), _p(v), _p(t))
    return q, v, t


def observation(P, task, qpos, qvel, target):
    P = np.ascontiguousarray(P, dtype=np.float64)
    t = task_id(task)
    obs = np.zeros(OBS_DIM[t])
    lib().jbo_observation(_p(P), t, _p(np.ascontiguousarray(qpos, dtype=np.float64)), _p(np.ascontiguousarray(qvel, dtype=np.float64)),
                          _p(np.ascontiguousarray(target, dtype=np.float64)), _p(obs))
    return obs


def reward(P, task, qpos, qvel, target):
    P = np.ascontiguousarray(P, dtype=np.float64)
    return lib().jbo_reward(_p(P), task_id(task), _p(np.ascontiguousarray(qpos, dtype=np.float64)),
                            _p(np.ascontiguousarray(qvel, dtype=np.float64)), _p(np.ascontiguousarray(target, dtype=np.float64)))


def reward_terms(P, qpos, qvel, target):
    out = np.zeros(4)
    lib().jbo_reward_terms(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)),
                           _p(np.ascontiguousarray(qvel, dtype=np.float64)), _p(np.ascontiguousarray(target, dtype=np.float64)), _p(out))
    return dict(P=out[0], H=out[1], V=out[2], U=out[3])


def tolerance(x, bounds=(0.0, 0.0), margin=0.0, value_at_margin=0.1, sigmoid="gaussian"):
    kind = dict(gaussian=0, cosine=1, linear=2)[sigmoid]
    return lib().jbo_tolerance(float(x), float(bounds[0]), float(bounds[1]), float(margin), float(value_at_margin), kind)


def pair_mpr(P, qpos, leg, tol=0.0, iters=0):
    """MuJoCo's narrow phase (MPR; tol / iters default to MuJoCo's 1e-6 / 50) of (mass ellipsoid, upper cylinder of `leg`):
    (touching, depth_or_gap, normal[3] ellipsoid -> cylinder, pos[3]).  The oracle's substep does NOT use it, see jb_oracle.c."""
    out = np.zeros(7)
    L = lib()
    L.jbo_pair_mpr.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
    hit = L.jbo_pair_mpr(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)), int(leg), float(tol), int(iters), _p(out))
    return bool(hit), out[0], out[1:4].copy(), out[4:7].copy()


def pair_geometric(P, qpos, leg, exact=False):
    """The geometric contact of the same pair, which collide() uses: (defined, dist (<0: penetration), normal[3], pos[3]).
    exact=True: the same contact by bracketed iterations run to convergence instead of the kernel's fixed scheme."""
    out = np.zeros(7)
    L = lib()
    fn = L.jbo_pair_geometric_exact if exact else L.jbo_pair_geometric
    fn.argtypes = [_dp, _dp, C.c_int, _dp]
    ok = fn(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)), int(leg), _p(out))
    return bool(ok), out[0], out[1:4].copy(), out[4:7].copy()


def pair_thread_geometric(P, qpos, leg, steps=0):
    """The simulated contact of the motor-axis thread (geom 20) with the upper-leg cylinder of `leg` (jb_oracle.c pair_thread_geometric):
    -> (dist, normal leg -> thread, position, distance of the two axis segments).  steps: bisection steps (0: the kernel's fixed count)."""
    out = np.zeros(8)
    L = lib()
    L.jbo_pair_thread_geometric.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp]
    L.jbo_pair_thread_geometric(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)), int(leg), int(steps), _p(out))
    return out[0], out[1:4].copy(), out[4:7].copy(), out[7]


def pair_clearance(P, qpos, skip_simulated=False):
    """Minimum distance over the geom pairs MuJoCo's filters would test (jb_clearance.c).  qpos [16] -> (distance, (gi, gj));
    qpos [n,16] -> (distances [n], pairs [n,2]); P may be one table or one per row.  skip_simulated: leave out the pairs the simulator
    collides itself (mass ellipsoid and motor-axis thread against the upper-leg cylinders)."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.ascontiguousarray(qpos, dtype=np.float64)
    one = q.ndim == 1
    q = np.atleast_2d(q)
    n = q.shape[0]
    out = np.zeros(n)
    pairs = np.zeros((n, 2), dtype=np.int32)
    L = lib()
    L.jbo_pair_clearance_batch2.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, C.c_void_p, C.c_int, C.c_int]
    L.jbo_pair_clearance_batch2(_p(P), int(P.ndim == 2), n, _p(q), _p(out), pairs.ctypes.data, usable_cores(), int(bool(skip_simulated)))
    if one:
        return float(out[0]), (int(pairs[0, 0]), int(pairs[0, 1]))
    return out, pairs


def mass_sweep_clearance(P, n_phi=72):
    """Rest pose, minimum over n_phi motor angles of the distance between the eccentric-mass body's geoms and every leg geom:
    how freely the mass can turn.  P: one table or [n, NPARAM] -> float or [n]."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    n = 1 if P.ndim == 1 else P.shape[0]
    out = np.zeros(n)
    lib().jbo_mass_sweep_clearance_batch(_p(P), int(P.ndim == 2), n, int(n_phi), _p(out))
    return out[0] if P.ndim == 1 else out


def geom_distance(P, qpos, gi, gj):
    return lib().jbo_geom_distance(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)), int(gi), int(gj))


def geom_world(P, qpos, gi):
    c, R, s = np.zeros(3), np.zeros(9), np.zeros(3)
    lib().jbo_geom_world(_p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(qpos, dtype=np.float64)), int(gi), _p(c), _p(R), _p(s))
    return c, R.reshape(3, 3), s


class OracleEnv:
    """Batch of N oracle environments in lockstep (fp64, OpenMP over envs)."""

    def __init__(self, n, task="move_from_origin", P=None, random_pose=True, nsub=50, step_limit=1000, seed=0,
                 env_offset=0, opts=None, per_env_model=False):
        self.n = int(n)
        self.task = task_id(task)
        self.D = OBS_DIM[self.task]
        P = np.ascontiguousarray(P, dtype=np.float64)
        self.opts = opts or default_opts()
        self._h = lib().jbo_env_create(self.n, self.task, int(random_pose), int(nsub), int(step_limit), seed, env_offset,
                                       _p(P), int(per_env_model), C.byref(self.opts))

    def __del__(self):
        try:
            if self._h:
                lib().jbo_env_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self, mask=None):
        obs = np.zeros((self.n, self.D))
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
        lib().jbo_env_reset(self._h, m.ctypes.data if m is not None else None, _p(obs))
        return obs

    def step(self, action, auto_reset=True, nthreads=0):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(action, dtype=np.float64).reshape(-1), (self.n,)), dtype=np.float64)
        obs = np.zeros((self.n, self.D))
        rew = np.zeros(self.n)
        done = np.zeros(self.n, dtype=np.uint8)
        lib().jbo_env_step(self._h, _p(a), _p(obs), _p(rew), done.ctypes.data, int(auto_reset), int(nthreads) if nthreads else usable_cores())
        return obs, rew, done

    def get_state(self):
        q, v, t = np.zeros((self.n, NQ)), np.zeros((self.n, NV)), np.zeros((self.n, 3))
        lib().jbo_env_get_state(self._h, _p(q), _p(v), _p(t))
        return q, v, t

    def set_state(self, qpos=None, qvel=None, target=None):
        def prep(a, w):
            return None if a is None else _p(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(self.n, w)))
        lib().jbo_env_set_state(self._h, prep(qpos, NQ), prep(qvel, NV), prep(target, 3))

    def counters(self):
        sc = np.zeros(self.n, dtype=np.int32)
        ep = np.zeros(self.n, dtype=np.uint32)
        lib().jbo_env_get_counters(self._h, sc.ctypes.data, ep.ctypes.data)
        return sc, ep

    def margins(self):
        """Per env: the smallest |distance| of any contact candidate point during the last control step (jbo_stats.margin_min)."""
        m = np.zeros(self.n)
        lib().jbo_env_get_margin(self._h, _p(m))
        return m

    def stats(self):
        s = Stats()
        lib().jbo_env_stats(self._h, C.byref(s))
        return s
