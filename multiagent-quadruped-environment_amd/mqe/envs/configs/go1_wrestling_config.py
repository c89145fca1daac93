"""go1wrestling: two robots on a raised ring (values: reference mqe/envs/configs/go1_wrestling_config.py)."""
from mqe.envs.configs._build import cfg

Go1WrestlingCfg = cfg("Go1WrestlingCfg")
