"""gym.spaces stand-ins: metadata containers only (rlskyjo/environment/skyjo_env.py:125-151)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete:
    def __init__(self, n):
        self.n = int(n)

    def contains(self, x):
        try:
            return 0 <= int(x) < self.n and int(x) == x
        except (TypeError, ValueError):
            return False


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]
