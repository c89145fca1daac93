"""Unitree Go1 defaults: asset, control, observation and termination switches (table entry `Go1Cfg`; values: reference
mqe/envs/go1/go1_config.py:9-247)."""
from mqe.envs.configs._build import cfg

Go1Cfg = cfg("Go1Cfg")
