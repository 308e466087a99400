"""Batched Jitterbug environment on one MI355X.

Host mirror of the reference's vectorised use of the env (stable-baselines VecEnv in
reference benchmarks/benchmark.py:146-185: reset() -> obs[N,D], step(actions) ->
(obs, reward, done, infos), auto-reset on done), backed by libjitterbug_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib, model, variants

TASKS = model.TASKS
DEFAULT_TIME_LIMIT = 10          # reference jitterbug.py:56
DEFAULT_CONTROL_TIMESTEP = 0.01  # reference jitterbug.py:57
PHYSICS_TIMESTEP = 0.0002        # reference jitterbug.xml:18


def compute_n_substeps(control_timestep, physics_timestep=PHYSICS_TIMESTEP, tol=1e-8):
    """dm_control.rl.control.compute_n_steps: control step must be an integer multiple of the physics step."""
    if control_timestep < physics_timestep:
        raise ValueError("Control timestep ({}) cannot be smaller than physics timestep ({}).".format(control_timestep, physics_timestep))
    n = control_timestep / physics_timestep
    if abs(n - round(n)) > tol:
        raise ValueError("Control timestep ({}) must be an integer multiple of physics timestep ({})".format(control_timestep, physics_timestep))
    return int(round(n))


class LazyInfos(list):
    """A list of N info dicts whose dicts come into being when an entry is first read (indexing, iteration, slicing)."""

    def __init__(self, n):
        super().__init__([None] * n)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        d = super().__getitem__(i)
        if d is None:
            d = {}
            super().__setitem__(i, d)
        return d

    def __iter__(self):
        for j in range(len(self)):
            yield self[j]

    def __eq__(self, other):
        return list(self) == list(other)

    __hash__ = None


class JitterbugVecEnv:
    """N lockstep Jitterbug environments on one GPU."""

    def __init__(self, n_envs, task="move_from_origin", seed=0, device_id=0, random_pose=True, contacts=True,
                 time_limit=DEFAULT_TIME_LIMIT, control_timestep=DEFAULT_CONTROL_TIMESTEP, auto_reset=True,
                 env_offset=0, max_newton=0, stream=None, params=None, envs_per_wave=0, flags=0, variant=None, per_env_model=False,
                 envs_per_gpu=None, pair_witness=False):
        """variant: 'auto' | 'ordinary' | 'lean' (jitterbug_amd.variants: 'auto' picks the two-waves-per-SIMD kernel above 4096 envs per
        GPU, from 8192 with one model per env); None keeps `flags` as given (JB_FLAG_LEAN = 2 by hand).  per_env_model: the batch will get
        one model per env (randomise_models / set_model_params with N tables) - 'auto' needs to know at creation.  envs_per_gpu: what
        'auto' is resolved from when this env is one shard of a larger batch (every shard must make the same choice); default n_envs."""
        if task not in TASKS:
            raise AssertionError("Invalid task {}, options are {}".format(task, list(TASKS)))   # reference jitterbug.py:425
        self._L = _lib.load()
        self.task = task
        self.task_id = TASKS.index(task)
        self.num_envs = int(n_envs)
        self.obs_dim = model.OBS_DIM[task]
        self.substeps = compute_n_substeps(control_timestep)
        self.control_timestep = float(control_timestep)
        if time_limit == float("inf"):
            self.step_limit = 2 ** 31 - 1
        else:
            self.step_limit = int(np.ceil(time_limit / (PHYSICS_TIMESTEP * self.substeps) - 1e-9))
        cfg = _lib.Config()
        _lib.check(self._L.jb_default_config(C.byref(cfg), self.num_envs, self.task_id))
        cfg.device_id = int(device_id)
        cfg.random_pose = int(bool(random_pose))
        cfg.contacts = int(bool(contacts))
        cfg.substeps = self.substeps
        cfg.step_limit = self.step_limit
        cfg.auto_reset = int(bool(auto_reset))
        cfg.max_newton = int(max_newton)
        cfg.envs_per_wave = int(envs_per_wave)
        if variant is not None:
            flags = variants.flags_for(variant, self.num_envs if envs_per_gpu is None else envs_per_gpu, per_env_model, flags)
        if pair_witness:
            flags = int(flags) | _lib.FLAG_PAIR_WITNESS
        cfg.flags = int(flags)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.env_offset = int(env_offset)
        cfg.use_caller_stream = 0 if stream is None else 1
        cfg.stream = None if stream is None else int(stream)
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self._L.jb_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._obs = np.zeros((self.num_envs, self.obs_dim), dtype=np.float32)
        self._rew = np.zeros(self.num_envs, dtype=np.float32)
        self._done = np.zeros(self.num_envs, dtype=np.uint8)
        self._pending = None
        self._views = None               # numpy views of the pinned result buffers of step_async / step_wait
        self._params = None              # host copy of the per-env / shared parameter table(s) when set_model_params was used
        self._rnd = None                 # randomise_models without fetched tables: what model_params() needs to rebuild one on the host
        self.state_version = 0           # bumped by every call that changes the simulator state (Physics caches on it)
        if params is not None:
            self.set_model_params(params)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.jb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ host-buffer API
    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _lib.check(self._L.jb_reset(self._h, _lib.ptr(m), _lib.ptr(self._obs)))
        self.state_version += 1
        return self._obs.copy()

    def step(self, actions):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(actions, dtype=np.float32).reshape(-1), (self.num_envs,)))
        _lib.check(self._L.jb_step(self._h, _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(self._rew), _lib.ptr(self._done)))
        self.state_version += 1
        return self._obs.copy(), self._rew.copy(), self._done.astype(bool), self._empty_infos()

    def _empty_infos(self):
        """VecEnv `infos`: always a list of N dicts (stable-baselines consumers index infos[i] and write keys such as 'episode' or
        'terminal_observation' into it).  Every step returns its OWN list, so nothing a wrapper writes survives into a later step;
        for large batches the dicts are made when first touched (65 536 dict allocations per step would cost more than the step)."""
        if self.num_envs <= 64:
            return [{} for _ in range(self.num_envs)]
        return LazyInfos(self.num_envs)

    def step_async(self, actions):
        """First half of step() (the VecEnv split the reference's SubprocVecEnv workers give its trainer, benchmarks/benchmark.py:146-171):
        the actions go into pinned staging, H2D copy -> step kernel -> D2H copies are queued on the handle's stream, and the call
        returns while the GPU works (jb_step_async)."""
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(actions, dtype=np.float32).reshape(-1), (self.num_envs,)))
        _lib.check(self._L.jb_step_async(self._h, _lib.ptr(a)))
        self._pending = True
        self.state_version += 1

    def step_wait(self, copy=True):
        """Second half: waits for the event behind the copies (jb_step_wait) and hands out (obs, reward, done, infos) - the very
        arrays step() would have returned.  copy=False: views of the handle's pinned buffers, valid until the next step_async."""
        if not self._pending:
            raise RuntimeError("step_wait() without a step_async() before it")
        self._pending = None
        if copy:          # straight into arrays of the caller's own (one copy out of the pinned buffers, none on top)
            o, r, d = np.empty((self.num_envs, self.obs_dim), dtype=np.float32), np.empty(self.num_envs, dtype=np.float32), np.empty(self.num_envs, dtype=np.uint8)
            _lib.check(self._L.jb_step_wait(self._h, _lib.ptr(o), _lib.ptr(r), _lib.ptr(d)))
            return o, r, d.view(np.bool_), self._empty_infos()
        _lib.check(self._L.jb_step_wait(self._h, None, None, None))
        if self._views is None:
            po, pr, pd = C.c_void_p(), C.c_void_p(), C.c_void_p()
            _lib.check(self._L.jb_step_views(self._h, C.byref(po), C.byref(pr), C.byref(pd)))
            n, d = self.num_envs, self.obs_dim
            self._views = (np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_float)), shape=(n, d)), np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_float)), shape=(n,)),
                           np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_uint8)), shape=(n,)))
        o, r, dn = self._views
        return o, r, dn.view(np.bool_), self._empty_infos()

    def release_staging(self):
        """Frees the device staging that host-buffer rollouts have grown (jb_release_staging): K x N x (D+2) floats stay with the handle otherwise."""
        _lib.check(self._L.jb_release_staging(self._h))

    def observe(self):
        _lib.check(self._L.jb_observe(self._h, _lib.ptr(self._obs), _lib.ptr(self._rew)))
        return self._obs.copy(), self._rew.copy()

    def get_state(self):
        q = np.zeros((self.num_envs, model.NQ))
        v = np.zeros((self.num_envs, model.NV))
        t = np.zeros((self.num_envs, 3))
        _lib.check(self._L.jb_get_state(self._h, _lib.ptr(q), _lib.ptr(v), _lib.ptr(t)))
        return q, v, t

    def set_state(self, qpos=None, qvel=None, target=None):
        def prep(a, w):
            return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(self.num_envs, w))
        q, v, t = prep(qpos, model.NQ), prep(qvel, model.NV), prep(target, 3)
        _lib.check(self._L.jb_set_state(self._h, _lib.ptr(q), _lib.ptr(v), _lib.ptr(t)))
        self.state_version += 1

    def counters(self):
        sc = np.zeros(self.num_envs, dtype=np.int32)
        ep = np.zeros(self.num_envs, dtype=np.uint32)
        cap = np.zeros(self.num_envs, dtype=np.float32)
        _lib.check(self._L.jb_get_counters(self._h, _lib.ptr(sc), _lib.ptr(ep), _lib.ptr(cap)))
        return sc, ep, cap

    def solver_stats(self):
        """Wave-substeps (since the env was created) whose contact problem went to the line-searched Newton solve because the plain
        active-set iteration had not settled within `max_newton` checks (jb_solver_stats)."""
        import ctypes
        v = ctypes.c_uint64(0)
        _lib.check(self._L.jb_solver_stats(self._h, ctypes.byref(v)))
        return int(v.value)

    def set_model_params(self, params):
        p = np.ascontiguousarray(params, dtype=np.float64)
        n_tables = 1 if p.ndim == 1 else p.shape[0]
        assert p.size == n_tables * model.NPARAM
        _lib.check(self._L.jb_set_model_params(self._h, _lib.ptr(p), n_tables))
        self._params = p.reshape(n_tables, model.NPARAM).copy()
        self._rnd = None
        self.state_version += 1

    def randomise_models(self, seed=0, modify_legs=True, modify_mass=True, modify_coreBody1=False, modify_coreBody2=False,
                         modify_global_density=False, modify_gear=False, sd_legs=None, sd_mass_pos=None, min_mass_clearance=0.0,
                         offsets=None, return_params=None, return_offsets=False):
        """One randomised model per env, generated ON THE GPU (jb_randomise_models; reference augment_Jitterbug's keyword arguments,
        augmented_jitterbug.py:95-107).  `offsets` [N, 31]: compile these instead of drawing.  Returns a dict with 'attempts' and,
        when asked for, 'params' [N, NPARAM] / 'offsets' [N, 31] (params are fetched by default up to 8192 envs: the Physics
        accessors of a per-env model read them)."""
        cfg = _lib.RandomiseConfig()
        _lib.check(self._L.jb_default_randomise_config(C.byref(cfg)))
        cfg.flags = ((_lib.RND_LEGS if modify_legs else 0) | (_lib.RND_MASS if modify_mass else 0) | (_lib.RND_CORE1_DENSITY if modify_coreBody1 else 0)
                     | (_lib.RND_CORE2_DENSITY if modify_coreBody2 else 0) | (_lib.RND_GLOBAL_DENSITY if modify_global_density else 0) | (_lib.RND_GEAR if modify_gear else 0))
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.min_mass_clearance = float(min_mass_clearance)
        if sd_legs is not None:
            cfg.sd_legs[:] = [float(x) for x in sd_legs]
        if sd_mass_pos is not None:
            cfg.sd_mass_pos[:] = [float(x) for x in sd_mass_pos]
        # the reference's sigmas are the tested distribution (tests/test_gpu_clearance.py: no unsimulated geom pair ever touches); beyond them
        # bodies may pass through each other where MuJoCo would collide them - the pair witness says whether they do
        ref_legs, ref_mass = (0.003, 0.003, 0.002), (0.0015, 0.002, 0.001)
        wide = any(float(x) > r * (1 + 1e-9) for x, r in zip(cfg.sd_legs, ref_legs)) or any(float(x) > r * (1 + 1e-9) for x, r in zip(cfg.sd_mass_pos, ref_mass))
        if wide and not (int(self.cfg.flags) & _lib.FLAG_PAIR_WITNESS):
            import warnings
            warnings.warn("randomise_models: sigmas beyond the reference's (augmented_jitterbug.py:96-98) - only the mass / thread against the upper legs are collided as geom-geom "
                          "pairs; create the env with pair_witness=True (JB_FLAG_PAIR_WITNESS) or call pair_witness() to see whether one of the other 152 pairs interpenetrates", RuntimeWarning)
        n = self.num_envs
        if return_params is None:
            return_params = n <= 8192
        off_in = None if offsets is None else np.ascontiguousarray(np.asarray(offsets, dtype=np.float64).reshape(n, _lib.NOFFSET))
        par = np.zeros((n, model.NPARAM)) if return_params else None
        off = np.zeros((n, _lib.NOFFSET)) if return_offsets else None
        att = np.zeros(n, dtype=np.int32)
        # (a failure raises here and leaves the handle - and self._params - on the model it had: jb_randomise_models swaps the new
        #  tables in only after the kernel succeeded)
        _lib.check(self._L.jb_randomise_models(self._h, C.byref(cfg), _lib.ptr(off_in), _lib.ptr(par), _lib.ptr(off), _lib.ptr(att)))
        self._params = par
        # tables not fetched (the default above 8192 envs, 5 KB each): model_params() rebuilds the ones it is asked for on the host
        # from the same Philox draws (the generator is keyed (seed, global env, attempt) and its compiler has a host twin)
        self._rnd = None if return_params else dict(cfg=cfg, attempts=att, offsets=off_in, cache={})
        self.state_version += 1
        out = dict(attempts=att)
        if return_params:
            out["params"] = par
        if return_offsets:
            out["offsets"] = off
        return out

    def pair_witness(self):
        """One pass of the pair witness on the current state (jb_pair_witness): (clearance [N] in metres - the smallest exact distance over the
        152 geom pairs MuJoCo would test and the simulator does not collide, 0 = a pair interpenetrates -, pairs [N, 2] geom indices)."""
        d = np.zeros(self.num_envs, dtype=np.float32)
        pr = np.zeros((self.num_envs, 2), dtype=np.int32)
        _lib.check(self._L.jb_pair_witness(self._h, _lib.ptr(d), pr.ctypes.data, None))
        return d, pr

    def pair_witness_counters(self):
        """Handles made with pair_witness=True (JB_FLAG_PAIR_WITNESS): (passes [N] that found an unsimulated pair interpenetrating - one pass
        runs behind every step launch -, smallest clearance [N] seen) since the env was created."""
        c = np.zeros(self.num_envs, dtype=np.uint32)
        m = np.zeros(self.num_envs, dtype=np.float32)
        _lib.check(self._L.jb_get_pair_witness(self._h, c.ctypes.data, _lib.ptr(m)))
        return c, m

    def model_params(self, index=0):
        """The parameter table (float64[NPARAM]) env `index` is simulated with."""
        if self._params is None:
            rnd = getattr(self, "_rnd", None)
            if rnd is None:
                return model.default_params()
            # per-env models whose tables stayed on the device: the host twin of the device generator, same draws, same compiler
            index = int(index)
            if index not in rnd["cache"]:
                if rnd["offsets"] is not None:
                    off = np.ascontiguousarray(rnd["offsets"][index], dtype=np.float64)
                else:
                    off = np.zeros(_lib.NOFFSET)
                    _lib.check(self._L.jb_model_draw_offsets_host(C.c_uint64(int(rnd["cfg"].seed)), C.c_uint64(int(self.cfg.env_offset) + index),
                                                                  C.c_uint32(int(rnd["attempts"][index]) - 1), C.byref(rnd["cfg"]), _lib.ptr(off)))
                P = np.zeros(model.NPARAM)
                _lib.check(self._L.jb_model_compile_host(_lib.ptr(off), int(rnd["cfg"].flags), _lib.ptr(P)))
                if len(rnd["cache"]) > 4096:
                    rnd["cache"].clear()
                rnd["cache"][index] = P
            return rnd["cache"][index]
        return self._params[index if self._params.shape[0] > 1 else 0]

    def reward_terms(self):
        """[N, 4] = position, heading, velocity, upright reward terms of the current state (reference jitterbug.py:840-889)."""
        out = np.zeros((self.num_envs, 4), dtype=np.float32)
        _lib.check(self._L.jb_reward_terms(self._h, _lib.ptr(out)))
        return out

    def set_policy_params(self, kick_angle=np.deg2rad(45), speed=0.3, angle_threshold=np.deg2rad(20)):
        """Keyword arguments of the reference's heuristic policies (heuristic_policies.py:28, 64, 81, 98) for policy() / rollouts."""
        _lib.check(self._L.jb_set_policy_params(self._h, float(kick_angle), float(speed), float(angle_threshold)))

    # ------------------------------------------------------------------ device-buffer API (raw pointers; torch tensors via .data_ptr())
    def step_device(self, action_ptr, obs_ptr, reward_ptr, done_ptr):
        _lib.check(self._L.jb_step_device(self._h, action_ptr, obs_ptr, reward_ptr, done_ptr))
        self.state_version += 1

    # ------------------------------------------------------------------ observation encoder hook (reference jitterbug.py:760-761, 927-993)
    def set_obs_encoder(self, layers, vae=False):
        """layers: list of (W [in, out], b [out], activation in {'linear', 'tanh', 'relu'}) (see jitterbug_amd.encoders); None / [] removes it."""
        acts = {"linear": 0, "tanh": 1, "relu": 2}
        if not layers:
            _lib.check(self._L.jb_set_obs_encoder(self._h, 0, None, None, None, None, 0))
            return
        dims = [int(np.asarray(layers[0][0]).shape[0])] + [int(np.asarray(W).shape[1]) for W, _, _ in layers]
        a = np.array([acts[act] for _, _, act in layers], dtype=np.int32)
        w = np.concatenate([np.asarray(W, dtype=np.float32).reshape(-1) for W, _, _ in layers])
        b = np.concatenate([np.asarray(bb, dtype=np.float32).reshape(-1) for _, bb, _ in layers])
        for (W, bb, _), din, dout in zip(layers, dims[:-1], dims[1:]):
            assert np.asarray(W).shape == (din, dout) and np.asarray(bb).shape == (dout,), "layer shapes must chain"
        d = np.array(dims, dtype=np.int32)
        _lib.check(self._L.jb_set_obs_encoder(self._h, len(layers), d.ctypes.data, a.ctypes.data, _lib.ptr(w), _lib.ptr(b), int(bool(vae))))

    @property
    def encoded_dim(self):
        return int(self._L.jb_encoded_dim(self._h))

    def encode(self, obs):
        """[N, D] observation rows -> [N, encoded_dim] codes (needs set_obs_encoder)."""
        o = np.ascontiguousarray(obs, dtype=np.float32).reshape(self.num_envs, self.obs_dim)
        n_out = self.encoded_dim
        out = np.empty((self.num_envs, max(n_out, 1)), dtype=np.float32)
        _lib.check(self._L.jb_encode(self._h, _lib.ptr(o), _lib.ptr(out)))
        return out

    def encode_device(self, obs_ptr, code_ptr):
        _lib.check(self._L.jb_encode_device(self._h, obs_ptr, code_ptr))

    def debug_poison_lds(self):
        """Diagnostic: fill the GPU's LDS with NaN patterns (see jb_debug_poison_lds)."""
        _lib.check(self._L.jb_debug_poison_lds(self._h))

    def step_rows_device(self, action_ptr, rows_ptr):
        """One packed float row [obs(D) | reward | done] per env, written by the step kernel (the unit of the multi-GPU gather)."""
        _lib.check(self._L.jb_step_rows_device(self._h, action_ptr, rows_ptr))
        self.state_version += 1

    def reset_device(self, mask_ptr=None, obs_ptr=None):
        _lib.check(self._L.jb_reset_device(self._h, mask_ptr, obs_ptr))
        self.state_version += 1

    def observe_device(self, obs_ptr, reward_ptr=None):
        _lib.check(self._L.jb_observe_device(self._h, obs_ptr, reward_ptr))

    def policy(self, obs):
        """The reference's heuristic policy for this task, evaluated on the GPU: obs [N,D] -> actions [N]."""
        o = np.ascontiguousarray(obs, dtype=np.float32).reshape(self.num_envs, self.obs_dim)
        a = np.zeros(self.num_envs, dtype=np.float32)
        _lib.check(self._L.jb_policy(self._h, _lib.ptr(o), _lib.ptr(a)))
        return a

    def policy_device(self, obs_ptr, action_ptr):
        _lib.check(self._L.jb_policy_device(self._h, obs_ptr, action_ptr))

    def rollout_policy(self, n_steps):
        """n_steps of heuristic policy -> step chained on the GPU from the current state; returns (rewards [n_steps, N], last obs)."""
        rew = np.zeros((int(n_steps), self.num_envs), dtype=np.float32)
        _lib.check(self._L.jb_rollout_policy(self._h, int(n_steps), _lib.ptr(rew), _lib.ptr(self._obs)))
        self.state_version += 1
        return rew, self._obs.copy()

    def rollout_policy_device(self, n_steps, obs_ptr, rewards_ptr=None, done_ptr=None):
        """n_steps of heuristic policy -> step, chained on the GPU (device pointers; asynchronous)."""
        _lib.check(self._L.jb_rollout_policy_device(self._h, int(n_steps), obs_ptr, rewards_ptr, done_ptr))
        self.state_version += 1

    def step_many_device(self, n_steps, actions_ptr=None, rows_ptr=None, rewards_ptr=None, obs_last_ptr=None, done_last_ptr=None):
        """n_steps control steps in ONE kernel launch (jb_step_many_device; asynchronous, device pointers).  actions_ptr: an action
        tape [n_steps, N], or None = the handle's heuristic policy evaluated in the kernel.  rows_ptr [n_steps, N, D+2] (packed rows of
        every step) OR rewards_ptr [n_steps, N] / obs_last_ptr [N, D] / done_last_ptr [N].  Bit-identical to n_steps single-step calls."""
        _lib.check(self._L.jb_step_many_device(self._h, int(n_steps), actions_ptr, rows_ptr, rewards_ptr, obs_last_ptr, done_last_ptr))
        self.state_version += 1

    def rollout(self, n_steps, actions=None):
        """n_steps control steps in ONE kernel launch, host arrays in and out (jb_step_many): `actions` [n_steps, N] float32, or None =
        the task's heuristic policy evaluated in the kernel (set_policy_params).  Returns (obs [K, N, D], reward [K, N], done [K, N]) -
        what n_steps calls of step() would have returned, bit for bit (auto-reset included: the row of a finished episode's last step
        holds the new episode's first observation)."""
        K = int(n_steps)
        a = None
        if actions is not None:
            a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(K, self.num_envs))
        rows = np.empty((K, self.num_envs, self.obs_dim + 2), dtype=np.float32)
        _lib.check(self._L.jb_step_many(self._h, K, _lib.ptr(a), _lib.ptr(rows)))
        self.state_version += 1
        return rows[..., :-2].copy(), rows[..., -2].copy(), rows[..., -1] > 0.5

    # ---- rows between GPUs through the library's own RCCL binding (jb_comm_*: no torch.distributed on the data path)
    @staticmethod
    def comm_unique_id():
        """128 bytes that rank 0 makes and hands to every rank by its own means (jb_comm_unique_id)."""
        import ctypes
        uid = (ctypes.c_char * 128)()
        _lib.check(_lib.load().jb_comm_unique_id(uid))
        return bytes(uid)

    def comm_init(self, n_ranks, rank, uid):
        """Collective over the ranks: one RCCL communicator for this handle (jb_comm_init)."""
        import ctypes
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(uid))
        _lib.check(self._L.jb_comm_init(self._h, int(n_ranks), int(rank), buf))

    def comm_destroy(self):
        _lib.check(self._L.jb_comm_destroy(self._h))

    def comm_set_shards(self, sizes):
        """The envs every rank holds (the same list on every rank): the exchanges then move blocks of the longest shard (jb_comm_set_shards)."""
        a = np.ascontiguousarray(sizes, dtype=np.int32)
        _lib.check(self._L.jb_comm_set_shards(self._h, a.ctypes.data))

    def scatter_actions_device(self, all_ptr, local_ptr, count, stream=None):
        """Rank 0's [n_ranks, count] action blocks to every rank's [count] (grouped ncclSend / ncclRecv, jb_scatter_actions_device)."""
        _lib.check(self._L.jb_scatter_actions_device(self._h, all_ptr, local_ptr, int(count), stream, 0 if stream is None else 1))

    def gather_rows_device(self, rows_ptr, all_ptr=None, stream=None):
        """This rank's packed rows [N, D+2] to rank 0's [n_ranks, N, D+2] (grouped ncclSend / ncclRecv); asynchronous on `stream` (a raw
        hipStream_t) or, when None, on the handle's stream."""
        _lib.check(self._L.jb_gather_rows_device(self._h, rows_ptr, all_ptr, stream, 0 if stream is None else 1))

    def gather_block_device(self, src_ptr, all_ptr, count, stream=None):
        """`count` floats of every rank to rank 0's [n_ranks, count] (a fused rollout's [K, N, D+2] block)."""
        _lib.check(self._L.jb_gather_block_device(self._h, src_ptr, all_ptr, int(count), stream, 0 if stream is None else 1))

    def wave_clocks(self):
        """Seconds each wave of the last step launch was alive (its load imbalance: mean against max)."""
        out = np.zeros(self.num_envs, dtype=np.float64)
        n = int(self._L.jb_wave_clocks(self._h, _lib.ptr(out), self.num_envs))
        if n < 0:
            _lib.check(n)
        return out[:n]

    @property
    def kernel_variant(self):
        """'ordinary' | 'pair' | 'lean' | 'lean_pair': the step kernel this handle launches (jb_kernel_variant)."""
        if not hasattr(self._L, "jb_kernel_variant"):          # (an A/B library of an older ABI)
            return "ordinary"
        return _lib.VARIANT_NAMES[int(self._L.jb_kernel_variant(self._h))]

    @property
    def envs_per_wave(self):
        return int(self._L.jb_envs_per_wave(self._h))

    def synchronize(self):
        _lib.check(self._L.jb_synchronize(self._h))

    @property
    def stream(self):
        return self._L.jb_stream(self._h)


class MonitoredVecEnv:
    """stable-baselines ``Monitor`` semantics for a whole batch (reference benchmarks/benchmark.py:162-169, 177-184 wraps every
    env in ``bench.Monitor``): per-env episode return / length / wall time are accumulated on the host and reported in
    ``infos[i]['episode'] = {'r', 'l', 't'}`` on the step an env finishes; optionally appended to a monitor CSV with the
    same header convention (``#{json}`` line, then ``r,l,t`` rows)."""

    def __init__(self, venv, filename=None):
        import json
        import time
        self.venv = venv
        self.num_envs = venv.num_envs
        self._t0 = time.time()
        self._ret = np.zeros(self.num_envs)
        self._len = np.zeros(self.num_envs, dtype=np.int64)
        self.episode_returns, self.episode_lengths, self.episode_times = [], [], []
        self._fh = None
        if filename is not None:
            if not filename.endswith("monitor.csv"):
                filename = filename + ".monitor.csv"
            self._fh = open(filename, "wt")
            self._fh.write("#%s\n" % json.dumps({"t_start": self._t0, "env_id": "jitterbug-%s" % venv.task}))
            self._fh.write("r,l,t\n")

    def reset(self, mask=None):
        obs = self.venv.reset(mask)
        sel = np.ones(self.num_envs, bool) if mask is None else np.asarray(mask, bool)
        self._ret[sel] = 0
        self._len[sel] = 0
        return obs

    def step(self, actions):
        return self._account(*self.venv.step(actions)[:3])

    def _account(self, obs, rew, done):
        import time
        self._ret += rew
        self._len += 1
        infos = [{} for _ in range(self.num_envs)]
        if done.any():
            t = round(time.time() - self._t0, 6)
            for i in np.nonzero(done)[0]:
                ep = {"r": round(float(self._ret[i]), 6), "l": int(self._len[i]), "t": t}
                infos[i]["episode"] = ep
                self.episode_returns.append(ep["r"]); self.episode_lengths.append(ep["l"]); self.episode_times.append(t)
                if self._fh:
                    self._fh.write("%s,%d,%s\n" % (ep["r"], ep["l"], ep["t"]))
            self._ret[done] = 0
            self._len[done] = 0
            if self._fh:
                self._fh.flush()
        return obs, rew, done, infos

    def step_async(self, actions):
        self.venv.step_async(actions)

    def step_wait(self):
        return self._account(*self.venv.step_wait()[:3])

    def close(self):
        if self._fh:
            self._fh.close()
        self.venv.close()
