"""Minimal stand-ins for dm_env / dm_control specs and gym spaces.

dm_control, dm_env and gym are not installed in this image, so the surface the reference
relies on from them (third party: dm_env.TimeStep/StepType, dm_env.specs.Array/BoundedArray, gym.spaces.Box/Dict)
is re-provided here with the same field names and semantics."""
import collections
import enum

import numpy as np


class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

    def first(self):
        return self is StepType.FIRST

    def mid(self):
        return self is StepType.MID

    def last(self):
        return self is StepType.LAST


class TimeStep(collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])):
    __slots__ = ()

    def first(self):
        return self.step_type == StepType.FIRST

    def mid(self):
        return self.step_type == StepType.MID

    def last(self):
        return self.step_type == StepType.LAST


class Array:
    def __init__(self, shape, dtype, name=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.name = name

    def __repr__(self):
        return "Array(shape={}, dtype={}, name={!r})".format(self.shape, self.dtype, self.name)

    def validate(self, value):
        value = np.asarray(value)
        if value.shape != self.shape:
            raise ValueError("Expected shape %r but found %r" % (self.shape, value.shape))
        return value

    def generate_value(self):
        return np.zeros(self.shape, dtype=self.dtype)


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self.minimum = np.broadcast_to(np.asarray(minimum, dtype=self.dtype), self.shape).copy()
        self.maximum = np.broadcast_to(np.asarray(maximum, dtype=self.dtype), self.shape).copy()

    def __repr__(self):
        return "BoundedArray(shape={}, dtype={}, name={!r}, minimum={}, maximum={})".format(
            self.shape, self.dtype, self.name, self.minimum, self.maximum)


class Box:
    """gym.spaces.Box stand-in."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

    def __repr__(self):
        return "Box{}".format(self.shape)


class Dict:
    """gym.spaces.Dict stand-in."""

    def __init__(self, spaces):
        self.spaces = collections.OrderedDict(spaces)

    def sample(self):
        return collections.OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def seed(self, seed=None):
        return [s.seed(seed) for s in self.spaces.values()]

    def __getitem__(self, k):
        return self.spaces[k]

    def __repr__(self):
        return "Dict({})".format(", ".join("{}:{}".format(k, v) for k, v in self.spaces.items()))
