"""go1bridge: two robots meet on a narrow bridge (values: reference mqe/envs/configs/go1_bridge_config.py)."""
from mqe.envs.configs._build import cfg

Go1BridgeCfg = cfg("Go1BridgeCfg")
