"""Minimal stand-ins for the four ``gym.spaces`` classes the reference API derives from (gym 0.12 is not a
dependency here).  Only what pypownet's environment module and its agents use is provided: ``shape``, ``n``,
``spaces`` (ordered), ``sample()`` and ``contains()``."""
from collections import OrderedDict

import numpy as np


class Space(object):
    shape = None
    dtype = None

    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class MultiBinary(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = (self.n,)
        self.dtype = np.int8

    def sample(self):
        return np.random.randint(low=0, high=2, size=self.n).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(((x == 0) | (x == 1)).all())


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return np.random.randint(self.n)

    def contains(self, x):
        return 0 <= int(x) < self.n


class Box(Space):
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
        self.low = np.full(self.shape, low, dtype=np.float64) if np.isscalar(low) else np.asarray(low)
        self.high = np.full(self.shape, high, dtype=np.float64) if np.isscalar(high) else np.asarray(high)
        self.dtype = np.dtype(dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return np.random.uniform(lo, hi, size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = spaces if isinstance(spaces, OrderedDict) else OrderedDict(spaces)

    def sample(self):
        return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def contains(self, x):
        return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())
