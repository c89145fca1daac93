"""go1sheep-easy / -hard: two dogs herd one / nine scripted sheep through a gate (values: reference
mqe/envs/configs/go1_sheep_config.py)."""
from mqe.envs.configs._build import cfg

SingleSheepCfg = cfg("SingleSheepCfg")
NineSheepCfg = cfg("NineSheepCfg")
