"""jitterbug_amd - MI355X-native, lockstep-vectorised Jitterbug environment.

Drop-in surface of the reference package ``jitterbug_dmc`` for its hot path:
    from jitterbug_amd import suite
    env = suite.load("jitterbug", "move_from_origin", task_kwargs=dict(random=0))
    ts = env.reset(); ts = env.step(0.8)
and, where the batch lives, ``JitterbugVecEnv(n_envs, task)``.
"""
__version__ = "0.1.0"

from . import suite                                 # noqa: F401
from .gym_wrapper import JitterbugGymEnv            # noqa: F401
from .jitterbug import Jitterbug                    # noqa: F401
from .vec_env import JitterbugVecEnv                # noqa: F401

suite.register_with_dm_control()
