"""Defaults shared by every task (table entry `LeggedRobotCfg`; values: reference mqe/envs/base/legged_robot_config.py:33-229)."""
from mqe.envs.configs._build import cfg

LeggedRobotCfg = cfg("LeggedRobotCfg")
