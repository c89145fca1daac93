"""Terrain classes by name (reference mqe/utils/terrain/__init__.py:3-13): `cfg.terrain.selected` picks one."""
import importlib

from .barrier_track import BarrierTrack  # noqa: F401

terrain_registry = dict(
    Terrain="mqe.utils.terrain.terrain:Terrain",
    BarrierTrack="mqe.utils.terrain.barrier_track:BarrierTrack",
    TerrainPerlin="mqe.utils.terrain.perlin:TerrainPerlin",
)


def get_terrain_cls(terrain_cls):
    module, class_name = terrain_registry[terrain_cls].rsplit(":", 1)
    return getattr(importlib.import_module(module), class_name)
