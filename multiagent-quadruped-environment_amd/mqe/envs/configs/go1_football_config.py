"""go1football-defender / -1vs1 / -2vs2 (values: reference mqe/envs/configs/go1_football_config.py)."""
from mqe.envs.configs._build import cfg

Go1FootballDefenderCfg = cfg("Go1FootballDefenderCfg")
Go1Football1vs1Cfg = cfg("Go1Football1vs1Cfg")
Go1Football2vs2Cfg = cfg("Go1Football2vs2Cfg")
