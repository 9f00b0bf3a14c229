"""Minimal Box / Dict spaces (gym 0.21 is not installed anywhere this runs).

Only what the reference touches: ``gym.spaces.Box(low, high, shape, dtype)`` and ``gym.spaces.Dict``
(simulation/controller/actuator.py:245, simulation/controller/sensor.py:21-52), with ``sample`` and
``contains`` so that SB3-style code keeps working.
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape).astype(self.dtype, copy=True) if np.isfinite(low).all() else np.full(self.shape, low)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape).astype(self.dtype, copy=True) if np.isfinite(high).all() else np.full(self.shape, high)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        if self.dtype.kind in "ui":
            return self._rng.integers(self.low, self.high.astype(np.int64) + 1, size=self.shape).astype(self.dtype)
        lo = np.where(np.isfinite(self.low), self.low, -1e6); hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return self._rng.uniform(lo, hi, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Dict(dict):
    def __init__(self, spaces):
        super().__init__(spaces)
        self.spaces = self

    def sample(self):
        return {k: s.sample() for k, s in self.items()}

    def contains(self, x):
        return all(k in x and s.contains(x[k]) for k, s in self.items())
