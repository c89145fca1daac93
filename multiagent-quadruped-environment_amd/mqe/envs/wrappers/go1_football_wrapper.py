"""go1football-defender task wrapper (reference mqe/envs/wrappers/go1_football_wrapper.py:8-91): only agents 0,1
are exposed; obs (N,2,20) = [id2, own pos+rpy, other's, ball pos rel. env origin, ball vel]."""
import torch

from .empty_wrapper import EmptyWrapper, FusedTaskWrapper
from .spaces import Box


class Go1FootballDefenderWrapper(FusedTaskWrapper):
    task = "football_defender"
    wrapper_agents = 2

    def _obs_dim(self):
        return 18 + self.num_agents


class Go1FootballGameWrapper(EmptyWrapper):
    """go1football-1vs1 / -2vs2 (reference go1_football_wrapper.py:93-156).  Upstream this wrapper is a stub: reset()
    and step() return `None` observations and an all-zero reward of shape (num_envs, 4) whatever the number of
    agents; the physics underneath (robots + free ball) is complete.  Mirrored as is; actions are clipped to +-1 and
    scaled by [2, 0.5, 0.5] before Go1.step (:139-140), here inside the fused step."""

    def __init__(self, env):
        super().__init__(env)
        self.observation_space = Box(low=-float("inf"), high=float("inf"), shape=(18 + self.num_agents,), dtype=float)
        self.action_space = Box(low=-1, high=1, shape=(3,), dtype=float)
        self.action_scale = torch.tensor([[[2, 0.5, 0.5]]], device=env.device).repeat(self.num_envs, self.num_agents, 1)
        self.reward_buffer = {"goal reward": 0, "step count": 0}
        self._zero_reward = torch.zeros(self.num_envs, 4, device=env.device)

    def reset(self):
        self.env.reset()
        return None

    def step(self, action):
        action = torch.clip(action.to(self.action_scale.device, torch.float32).reshape(self.num_envs, self.num_agents, 3), -1, 1)
        self.env.step_fused(action * self.action_scale)      # "plain" engine task: Go1.step re-clips to +-1 (go1.py:38)
        self.reward_buffer["step count"] += 1
        return None, self._zero_reward.clone(), self.env.reset_buf, self.env.extras
