"""OpenAI-Gym style wrapper (reference: jitterbug_dmc/gym_wrapper.py:6-39 over dm2gym.DMControlEnv, third party).

``JitterbugGymEnv(env)`` keeps the reference's constructor and attributes (``num_envs``, ``render_every``) and the
dm2gym behaviour: ``reset() -> obs``, ``step(a) -> (obs, reward, done, info)``, ``action_space`` / ``observation_space``
converted from the env's specs (Box float32; Dict for dict observations).  Rendering (OpenGL/opencv in the reference) is
outside the hot path: ``render`` only counts frames."""
import numpy as np

from . import specs


def _convert_spec(spec):
    if isinstance(spec, dict):
        return specs.Dict([(k, _convert_spec(v)) for k, v in spec.items()])
    if isinstance(spec, specs.BoundedArray):
        return specs.Box(spec.minimum.astype(np.float32), spec.maximum.astype(np.float32), shape=spec.shape, dtype=np.float32)
    return specs.Box(-np.inf, np.inf, shape=spec.shape, dtype=np.float32)


class JitterbugGymEnv:
    metadata = {"render.modes": ["human", "rgb_array"]}
    reward_range = (0.0, 1.0)

    def __init__(self, env, *, render_every=1):
        self.num_envs = 1                      # reference gym_wrapper.py:18
        self.frame_count = 0
        self.render_every = render_every
        self.env = env
        self.action_space = _convert_spec(env.action_spec())
        self.observation_space = _convert_spec(env.observation_spec())
        self.timestep = None

    def seed(self, seed=None):
        self.action_space.seed(seed)
        return [seed]

    def reset(self):
        self.timestep = self.env.reset()
        return self.timestep.observation

    def step(self, action):
        self.timestep = self.env.step(action)
        return self.timestep.observation, self.timestep.reward, self.timestep.last(), {}

    def render(self, mode="human", **kwargs):
        self.frame_count += 1              # reference gym_wrapper.py:26; drawing itself is out of scope
        return None

    def close(self):
        self.env.close()
