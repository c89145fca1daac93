"""go1revolvingdoor: two robots pass a revolving door (values: reference mqe/envs/configs/go1_rotation_config.py)."""
from mqe.envs.configs._build import cfg

Go1RotationCfg = cfg("Go1RotationCfg")
