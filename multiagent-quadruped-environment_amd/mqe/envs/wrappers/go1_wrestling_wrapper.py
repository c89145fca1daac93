"""go1wrestling task wrapper (reference mqe/envs/wrappers/go1_wrestling_wrapper.py:9-89).  obs (N,A,12) = [own pos+rpy,
other's pos+rpy], agent 1's copy mirrored in y (entries 1, 4, 7, 10 negated); agent 1's y / yaw commands negated IN PLACE
(:48); reward (N,A,1), agent 0 only: +10 while the opponent is tipped over (|pitch| > 0.9 pi or |roll| >= 0.4 pi of
`base_quat`), -1 while agent 0 is."""
from .empty_wrapper import FusedTaskWrapper


class Go1WrestlingWrapper(FusedTaskWrapper):
    task = "wrestling"

    def _obs_dim(self):
        return 12

    def step(self, action):
        action[:, 1, 1:] = -action[:, 1, 1:]
        obs, rew, done, info = super().step(action)
        return obs, rew.reshape(self.num_envs, self.num_agents, 1), done, info
