"""`LeggedRobot` (reference mqe/envs/base/legged_robot.py:53-1172): step loop, resets, PD torques, termination.
All of it executes inside the HIP engine; the import path is preserved."""
from mqe.envs.go1.go1 import Go1 as LeggedRobot  # noqa: F401
