"""`BaseTask` of the reference (mqe/envs/base/base_task.py:38-150) allocates buffers and creates the Isaac Gym
sim/viewer.  Here the engine owns buffers and there is no viewer, so the name is kept as an alias of the one
concrete environment class for `isinstance` checks and imports."""
from mqe.envs.go1.go1 import Go1 as BaseTask  # noqa: F401
