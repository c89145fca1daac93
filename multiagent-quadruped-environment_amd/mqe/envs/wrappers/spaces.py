"""Minimal stand-ins for gym.spaces.Box / gym.Wrapper (gym is not a dependency of this build)."""


class Box:
    def __init__(self, low, high, shape=None, dtype=float):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape})"


class Wrapper:
    """Attribute access falls through to the wrapped env, like gym.Wrapper.__getattr__."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)
