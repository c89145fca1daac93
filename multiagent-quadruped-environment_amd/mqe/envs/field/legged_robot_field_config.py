"""BarrierTrack field defaults (table entry `LeggedRobotFieldCfg`; values: reference mqe/envs/field/legged_robot_field_config.py)."""
from mqe.envs.configs._build import cfg

LeggedRobotFieldCfg = cfg("LeggedRobotFieldCfg")
