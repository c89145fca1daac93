from .barrier_track import BarrierTrack  # noqa: F401


def get_terrain_cls(name):
    """Reference mqe/utils/terrain/__init__.py:3-13 resolves a class by name; only BarrierTrack is on the hot path."""
    if name == "BarrierTrack":
        return BarrierTrack
    raise NotImplementedError(f"terrain '{name}' is out of scope of this build (SURVEY.md section 2)")
