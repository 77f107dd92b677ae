"""Observation / action space descriptors (rlskyjo/environment/skyjo_env.py:125-151).

gym==0.21.0 is a pinned dependency of the reference that is not part of this image; when gym or
gymnasium is importable their classes are used, otherwise these metadata-only equivalents.
"""
import numpy as np

try:  # pragma: no cover - depends on the host image
    from gymnasium.spaces import Box, Dict, Discrete  # type: ignore
except Exception:  # noqa: BLE001
    try:
        from gym.spaces import Box, Dict, Discrete  # type: ignore
    except Exception:  # noqa: BLE001

        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

        class Discrete:
            def __init__(self, n):
                self.n = int(n)
                self.shape = ()
                self.dtype = np.dtype(np.int64)

            def contains(self, x):
                if isinstance(x, (bool, np.bool_)) or x is None:
                    return False
                try:
                    return int(x) == x and 0 <= int(x) < self.n
                except (TypeError, ValueError):
                    return False

            def __repr__(self):
                return f"Discrete({self.n})"

        class Dict:
            def __init__(self, spaces):
                self.spaces = dict(spaces)

            def __getitem__(self, key):
                return self.spaces[key]

            def contains(self, x):
                return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

            def __repr__(self):
                return f"Dict({self.spaces})"
