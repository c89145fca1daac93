"""Name-only stand-in for isaacgym.gymapi (import stub for golden generation)."""
SIM_PHYSX = 1
SIM_FLEX = 0
DOMAIN_SIM = 2
FOLLOW_TRANSFORM = 1
IMAGE_DEPTH = 1
IMAGE_COLOR = 0
KEY_ESCAPE = 0
KEY_V = 1


class _Bag:
    def __init__(self, *a, **k):
        self.__dict__.update(k)

    def __getattr__(self, name):  # lazily create nested bags (transform.p.x = ...)
        if name.startswith("__"):
            raise AttributeError(name)
        v = _Bag()
        object.__setattr__(self, name, v)
        return v


class Vec3(_Bag):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)


class Quat(_Bag):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Transform(_Bag):
    def __init__(self, p=None, r=None):
        self.p = p if p is not None else Vec3()
        self.r = r if r is not None else Quat()


class TriangleMeshParams(_Bag):
    def __init__(self):
        self.transform = Transform()


class PlaneParams(_Bag):
    pass


class HeightFieldParams(_Bag):
    def __init__(self):
        self.transform = Transform()


class AssetOptions(_Bag):
    pass


class CameraProperties(_Bag):
    pass


class SimParams(_Bag):
    pass


def acquire_gym():
    raise RuntimeError("isaacgym stub: no simulator")
