"""go1pushbox: two robots push a 6 kg box through a gate (values: reference mqe/envs/configs/go1_pushbox_config.py)."""
from mqe.envs.configs._build import cfg

Go1PushboxCfg = cfg("Go1PushboxCfg")
