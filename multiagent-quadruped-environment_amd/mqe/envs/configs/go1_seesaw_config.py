"""go1seesaw: two robots climb a seesaw onto a platform (values: reference mqe/envs/configs/go1_seesaw_config.py)."""
from mqe.envs.configs._build import cfg

Go1SeesawCfg = cfg("Go1SeesawCfg")
