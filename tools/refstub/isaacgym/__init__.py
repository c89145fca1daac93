from . import gymapi, gymutil, gymtorch, terrain_utils, torch_utils  # noqa: F401
