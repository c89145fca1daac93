"""go1tug: two robots on opposite sides of a 2.4 m disc that slides along y; each pushes it towards the other (values:
reference mqe/envs/configs/go1_tug_config.py:5-124; the start quaternions are unnormalised upstream, kept verbatim)."""
from mqe.envs.configs._build import cfg

Go1TugCfg = cfg("Go1TugCfg")
