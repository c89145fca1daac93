"""go1gate: two robots pass a narrow gate (values: reference mqe/envs/configs/go1_gate_config.py)."""
from mqe.envs.configs._build import cfg

Go1GateCfg = cfg("Go1GateCfg")
