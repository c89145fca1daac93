"""go1plane: one robot on the empty plane (values: reference mqe/envs/configs/go1_plane_config.py)."""
from mqe.envs.configs._build import cfg

Go1PlaneCfg = cfg("Go1PlaneCfg")
