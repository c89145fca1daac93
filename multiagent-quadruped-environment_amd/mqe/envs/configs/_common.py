"""Helpers shared by the task configs."""
from mqe.envs.go1.go1_config import Go1Cfg


def two_agents_at_origin(n=2):
    S = Go1Cfg.init_state
    return [S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]) for _ in range(n)]


def state(pos, rot=(0.0, 0.0, 0.0, 1.0)):
    return Go1Cfg.init_state(pos=list(pos), rot=list(rot), lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])
