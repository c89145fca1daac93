"""`LeggedRobotField` (reference mqe/envs/field/legged_robot_field.py:13-373): BarrierTrack hookup, roll/pitch/z
terminations, torque-limit override -- engine-side here; import path preserved."""
from mqe.envs.go1.go1 import Go1 as LeggedRobotField  # noqa: F401
