"""Host-side mirror of the reference's task module (reference: jitterbug_dmc/jitterbug.py).

Same names and call conventions as the reference for the hot path's surface:
  * five task factories ``move_from_origin ... move_to_pose`` returning an ``Environment``      (reference :72-174)
  * ``Physics`` accessors over the simulator state                                             (reference :177-317)
  * ``Jitterbug`` task: ``get_observation`` / ``get_reward`` / ``initialize_episode``          (reference :320-925)
  * ``Environment.reset/step/action_spec/observation_spec/physics/task/control_timestep``      (dm_control control.Environment,
    constructed at reference :84-90)
All arithmetic of a step (50 substeps, reward, observation, episode reset) runs in the HIP library through the C ABI;
this module only shapes its results into the reference's Python types.  The observation encoders the reference hard-wires
at HEAD (use_VAE=True, reference :468-477, 760-761) are NOT applied by default: observations are the raw dict documented in the
reference README (SURVEY.md finding 3).  `Environment(obs_encoder=layers)` switches the hook on with caller-supplied weights
(jitterbug_amd/encoders.py; the reference's weight files are not in its repository)."""
import collections

import numpy as np

from . import model, specs
from .vec_env import DEFAULT_CONTROL_TIMESTEP, DEFAULT_TIME_LIMIT, PHYSICS_TIMESTEP, JitterbugVecEnv

TARGET_SPEED = 0.1                       # reference jitterbug.py:58
TASK_NAMES = list(model.TASKS)

# observation dict layout per task: (key, width) in dict order (reference jitterbug.py:676-753)
_OBS_LAYOUT = {
    "move_from_origin": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1)],
    "face_direction": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("angle_to_target", 1)],
    "move_in_direction": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("angle_to_target", 1),
                          ("speed_in_target_frame", 3)],
    "move_to_position": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("target_in_jitterbug_frame", 3)],
    "move_to_pose": [("position", 7), ("velocity", 6), ("motor_position", 1), ("motor_velocity", 1), ("target_in_jitterbug_frame", 3),
                     ("angle_to_target", 1)],
}


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _wrap(angle):
    """(-pi, pi]  (reference jitterbug.py:235-238, 313-316)"""
    a = (angle + np.pi) % (2 * np.pi) - np.pi
    return np.pi if a == -np.pi else a


class _Indexer:
    """`physics.named.data.qpos['root']`-style access (dm_control's named indexing, third party) for the names the
    reference's hot path uses (reference jitterbug.py:182-296, 886)."""

    def __init__(self, getter):
        self._get = getter

    def __getitem__(self, key):
        return self._get(key)


class _NamedData:
    def __init__(self, ph):
        def qpos(k):
            q = ph._state()[0]
            return {"root": q[0:7], "jointMass": q[15:16]}[k] if isinstance(k, str) else q[k]

        def qvel(k):
            v = ph._state()[1]
            return {"root": v[0:6], "jointMass": v[14:15]}[k] if isinstance(k, str) else v[k]

        def xmat(k):
            body, comp = k if isinstance(k, tuple) else (k, None)
            R = _quat2mat(ph._state()[0][3:7]) if body == "jitterbug" else _quat2mat(ph.target_position_quat())
            return R.reshape(-1) if comp is None else R["xyz".index(comp[0]), "xyz".index(comp[1])]

        self.qpos, self.qvel, self.xmat = _Indexer(qpos), _Indexer(qvel), _Indexer(xmat)
        self.geom_xpos = _Indexer(lambda k: {"target": ph.target_position_xyz()}[k])
        self.xquat = _Indexer(lambda k: {"target": ph.target_position_quat(), "jitterbug": ph._state()[0][3:7]}[k])
        self.xpos = _Indexer(lambda k: {"target": ph.target_position_xyz(), "jitterbug": ph._state()[0][0:3]}[k])
        self.sensordata = _Indexer(lambda k: {"jitterbug_framelinvel": ph.jitterbug_framelinvel()}[k])


class _Named:
    def __init__(self, ph):
        self.data = _NamedData(ph)


class Physics:
    """Accessors of the reference's Physics class (reference jitterbug.py:177-317), evaluated on the state held by the GPU env
    (env `index` of the batch).  The state is fetched from the device once per simulator change (the env bumps
    `state_version` on every reset / step / set_state) and cached, so a reward assembled from a dozen accessors costs one
    device round trip, not a dozen."""

    def __init__(self, venv, index=0):
        self._venv = venv
        self._i = index
        self._cache = None
        self._cache_version = -1
        self.named = _Named(self)

    def _state(self):
        if self._cache is None or self._cache_version != self._venv.state_version:
            q, v, t = self._venv.get_state()
            self._cache = (q[self._i], v[self._i], t[self._i])
            self._cache_version = self._venv.state_version
        return self._cache

    @property
    def _params(self):
        return self._venv.model_params(self._i)

    @property
    def _target_z(self):
        return float(self._params[model.P_TARGETZ])

    @property
    def _root_ipos(self):
        """the root body's own centre of mass in root coordinates (MuJoCo body_ipos)"""
        return self._params[model.P_BODY + model.B_COM:model.P_BODY + model.B_COM + 3]

    # --- state vectors (MuJoCo layout) ---------------------------------------------------------
    def qpos(self):
        return self._state()[0]

    def qvel(self):
        return self._state()[1]

    def get_state(self):
        return np.concatenate(self._state()[:2])

    def timestep(self):
        return PHYSICS_TIMESTEP

    def time(self):
        sc, _, _ = self._venv.counters()
        return float(sc[self._i]) * self._venv.control_timestep

    # --- reference accessors -------------------------------------------------------------------
    def jitterbug_position(self):                      # :180
        return self._state()[0][0:7]

    def jitterbug_position_xyz(self):                  # :184
        return self.jitterbug_position()[:3]

    def jitterbug_position_quat(self):                 # :188
        return self.jitterbug_position()[3:]

    def jitterbug_direction_yaw(self):                 # :192-208
        R = _quat2mat(self.jitterbug_position_quat())
        return np.arctan2(R[1, 0], R[0, 0]) - np.pi / 2

    def jitterbug_velocity(self):                      # :210
        return self._state()[1][0:6]

    def jitterbug_velocity_xyz(self):                  # :214
        return self.jitterbug_velocity()[:3]

    def jitterbug_velocity_rpy(self):                  # :218
        return self.jitterbug_velocity()[3:]

    def motor_position(self):                          # :222-239
        return _wrap(self._state()[0][15] + np.pi / 2)

    def motor_velocity(self):                          # :241
        return self._state()[1][14]

    def target_position_xyz(self):                     # :254
        t = self._state()[2]
        return np.array([t[0], t[1], self._target_z])

    def target_position_quat(self):                    # :258
        psi = self._state()[2][2]
        return np.array([np.cos(psi / 2), 0.0, 0.0, np.sin(psi / 2)])

    def target_position(self):                         # :245-252
        return np.concatenate((self.target_position_xyz(), self.target_position_quat()), axis=0)

    def target_direction_yaw(self):                    # :262-273
        R = _quat2mat(self.target_position_quat())
        return np.arctan2(R[1, 0], R[0, 0])

    def target_position_in_jitterbug_frame(self):      # :275-290
        q, _, _ = self._state()
        return _quat2mat(q[3:7]).T @ (self.target_position_xyz() - q[:3])

    def jitterbug_framelinvel(self):
        """sensordata['jitterbug_framelinvel'] (reference jitterbug.xml:121, objtype="body"): MuJoCo measures a body object at
        its inertial frame, i.e. at the root body's own centre of mass, in world axes:  v + R (w_body x ipos)."""
        q, v, _ = self._state()
        return v[0:3] + _quat2mat(q[3:7]) @ np.cross(v[3:6], self._root_ipos)

    def jitterbug_velocity_in_target_frame(self):      # :292-303
        return _quat2mat(self.target_position_quat()).T @ self.jitterbug_framelinvel()

    def angle_jitterbug_to_target(self):               # :305-317
        return np.array([_wrap(self.target_direction_yaw() - self.jitterbug_direction_yaw())])

    def upright(self):
        """xmat['jitterbug','zz'] (reference :886)"""
        return _quat2mat(self.jitterbug_position_quat())[2, 2]


class Jitterbug:
    """The reference's task object: configuration + observation / reward / episode-start access (reference jitterbug.py:320-925).
    The methods that take `physics` evaluate on the GPU env behind it (jb_observe, jb_reward_terms, jb_reset through the C ABI);
    they exist so that code written against the reference's task API (`env.task.get_observation(env.physics)`, ...) runs."""

    # Approximate min / max ranges of the observation entries (reference jitterbug.py:324-347; the GPU observation kernel
    # applies the same numbers; pinned against the reference's literals by tests/golden/norm_tables.json)
    _NORM_ALL = np.array([[-2.0, 2.0], [-2.0, 2.0], [0.0, 0.1], [-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0],
                          [-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0], [-35.0, 35.0], [-35.0, 35.0], [-35.0, 35.0],
                          [-np.pi, np.pi], [-180.0, 180.0]])
    _NORM_TASKS = dict(                                         # reference :350-372
        move_from_origin=np.array([]),
        face_direction=np.array([[-np.pi, np.pi]]),
        move_in_direction=np.array([[-np.pi, np.pi], [-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0]]),
        move_to_position=np.array([[-3.0, 3.0], [-3.0, 3.0], [-0.1, 0.1]]),
        move_to_pose=np.array([[-3.0, 3.0], [-3.0, 3.0], [-0.1, 0.1], [-np.pi, np.pi]]))

    def __init__(self, random=None, task="move_from_origin", random_pose=True, norm_obs=False):
        assert task in TASK_NAMES, "Invalid task {}, options are {}".format(task, TASK_NAMES)      # :425
        self.task = task
        self.task_names = list(TASK_NAMES)
        self.random_pose = random_pose
        self.norm_obs = norm_obs            # accepted and ignored, like the reference (:430, 677-698)
        self.random = random
        self.counter = 0

    @staticmethod
    def _norm(v, min, max):                 # :668-671
        return (v - min) / (max - min) * 2.0 - 1.0

    # --- reference task API --------------------------------------------------------------------------
    def initialize_episode(self, physics):  # :601-666
        """Starts a new episode of the env behind `physics`: target and (random_pose) root orientation are re-drawn on the GPU
        from the env's Philox stream (one counter-based stream instead of the reference's mixed RNGs)."""
        mask = np.zeros(physics._venv.num_envs, dtype=np.uint8)
        mask[physics._i] = 1
        physics._venv.reset(mask)

    def get_observation(self, physics):     # :673-763
        """The (normalised) observation dict of the env behind `physics`, computed by the observation kernel."""
        obs, _ = physics._venv.observe()
        self.counter += 1                   # :758
        return self.split_observation(obs[physics._i])

    def _terms(self, physics):
        return physics._venv.reward_terms()[physics._i].astype(np.float64)

    def position_reward(self, physics):     # :868-880
        return float(self._terms(physics)[0])

    def heading_reward(self, physics):      # :840-852
        return float(self._terms(physics)[1])

    def velocity_reward(self, physics):     # :854-866
        return float(self._terms(physics)[2])

    def upright_reward(self, physics):      # :882-889
        return float(self._terms(physics)[3])

    def get_reward(self, physics):          # :891-925
        if physics._venv.task != self.task:
            raise ValueError("physics belongs to task {!r}, not {!r}".format(physics._venv.task, self.task))
        _, rew = physics._venv.observe()
        return float(rew[physics._i])

    def split_observation(self, vec):
        out = collections.OrderedDict()
        k = 0
        for name, w in _OBS_LAYOUT[self.task]:
            out[name] = np.asarray(vec[k:k + w], dtype=np.float64)
            k += w
        return out

    def obsdict2vec(self, obs):             # :765-838
        cols = {"move_from_origin": [], "face_direction": ["TargetYaw"],
                "move_in_direction": ["TargetYaw", "TargetVelX", "TargetVelY", "TargetVelZ"],
                "move_to_position": ["TargetX", "TargetY", "TargetZ"],
                "move_to_pose": ["TargetX", "TargetY", "TargetZ", "TargetYaw"]}[self.task]
        columns = ["X", "Y", "Z", "QuatX", "QuatY", "QuatZ", "QuatW", "VelX", "VelY", "VelZ", "VelRoll", "VelPitch", "VelYaw",
                   "MotorYaw", "MotorVelYaw"] + cols
        return np.concatenate([obs[name] for name, _ in _OBS_LAYOUT[self.task]]), columns


def _seed_from(random):
    if random is None:
        return int(np.random.randint(0, 2 ** 31 - 1))
    if isinstance(random, np.random.RandomState):
        return int(random.randint(0, 2 ** 31 - 1))
    return int(random)


class Environment:
    """dm_control ``control.Environment`` for one Jitterbug (third party; built at reference jitterbug.py:84-90)."""

    def __init__(self, task, time_limit=DEFAULT_TIME_LIMIT, control_timestep=DEFAULT_CONTROL_TIMESTEP, flat_observation=False,
                 device_id=0, obs_encoder=None, obs_encoder_vae=False, **unused_environment_kwargs):
        self._task = task
        self._flat_observation = flat_observation
        self._obs_encoder = obs_encoder       # layer list (jitterbug_amd.encoders): observations become {'observations': code},
                                              # what the reference's encode_obs returns (jitterbug.py:927-993).  Default: raw dict.
        self._venv = JitterbugVecEnv(1, task.task, seed=_seed_from(task.random), device_id=device_id, random_pose=task.random_pose,
                                     time_limit=time_limit, control_timestep=control_timestep, auto_reset=False)
        if obs_encoder:
            self._venv.set_obs_encoder(obs_encoder, vae=obs_encoder_vae)
        self._physics = Physics(self._venv)
        self._step_limit = float("inf") if time_limit == float("inf") else time_limit / (PHYSICS_TIMESTEP * self._venv.substeps)
        self._step_count = 0
        self._reset_next_step = True
        self._n_sub_steps = self._venv.substeps

    # --- dm_control surface ------------------------------------------------------------------------
    @property
    def physics(self):
        return self._physics

    @property
    def task(self):
        return self._task

    def control_timestep(self):
        return self._venv.control_timestep

    def action_spec(self):
        return specs.BoundedArray(shape=(1,), dtype=np.float64, minimum=-1.0, maximum=1.0)     # ctrlrange, reference jitterbug.xml:132-133

    def observation_spec(self):
        if self._obs_encoder:
            return collections.OrderedDict(observations=specs.Array((self._venv.encoded_dim,), np.float64, name="observations"))
        if self._flat_observation:
            return collections.OrderedDict(observations=specs.Array((self._venv.obs_dim,), np.float64, name="observations"))
        return collections.OrderedDict((name, specs.Array((w,), np.float64, name=name)) for name, w in _OBS_LAYOUT[self._task.task])

    def _observation(self, vec):
        if self._obs_encoder:
            return collections.OrderedDict(observations=self._venv.encode(vec[None, :])[0].astype(np.float64))
        return self._observation_raw(vec)

    def _observation_raw(self, vec):
        self._task.counter += 1
        vec = np.asarray(vec, dtype=np.float64)
        if self._flat_observation:
            return collections.OrderedDict(observations=vec)
        return self._task.split_observation(vec)

    def reset(self):
        self._reset_next_step = False
        self._step_count = 0
        obs = self._venv.reset()
        return specs.TimeStep(specs.StepType.FIRST, None, None, self._observation(obs[0]))

    def step(self, action):
        if self._reset_next_step:
            return self.reset()
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        obs, rew, done, _ = self._venv.step(np.array([a[0]], dtype=np.float32))
        self._step_count += 1
        if self._step_count >= self._step_limit:
            self._reset_next_step = True
            return specs.TimeStep(specs.StepType.LAST, float(rew[0]), 1.0, self._observation(obs[0]))
        return specs.TimeStep(specs.StepType.MID, float(rew[0]), 1.0, self._observation(obs[0]))

    def close(self):
        self._venv.close()


class _TaggedTasks(collections.OrderedDict):
    """dm_control containers.TaggedTasks stand-in: name -> factory, with tags."""

    def __init__(self):
        super().__init__()
        self._tags = collections.defaultdict(list)

    def add(self, *tags):
        def wrap(fn):
            self[fn.__name__] = fn
            for t in tags:
                self._tags[t].append(fn.__name__)
            return fn
        return wrap

    def tagged(self, tag):
        return list(self._tags[tag])


SUITE = _TaggedTasks()


def _factory(task_name):
    def make(time_limit=DEFAULT_TIME_LIMIT, control_timestep=DEFAULT_CONTROL_TIMESTEP, random=None, environment_kwargs=None, **kwargs):
        task = Jitterbug(random=random, task=task_name, **kwargs)
        environment_kwargs = environment_kwargs or {}
        return Environment(task, time_limit=time_limit, control_timestep=control_timestep, **environment_kwargs)
    make.__name__ = task_name
    make.__doc__ = "Jitterbug task %s (reference jitterbug.py:72-174)" % task_name
    return make


move_from_origin = SUITE.add("benchmarking", "easy")(_factory("move_from_origin"))
face_direction = SUITE.add("benchmarking", "easy")(_factory("face_direction"))
move_in_direction = SUITE.add("benchmarking", "easy")(_factory("move_in_direction"))
move_to_position = SUITE.add("benchmarking", "hard")(_factory("move_to_position"))
move_to_pose = SUITE.add("benchmarking", "hard")(_factory("move_to_pose"))
