"""go1revolvingdoor task wrapper (reference mqe/envs/wrappers/go1_rotation_wrapper.py:8-93).  obs (N,A,12) = [own pos+rpy,
other's pos+rpy] without the one-hot id, agent 1's copy mirrored in y (entries 1, 4, 7, 10 negated); agent 1's y / yaw
commands are negated IN PLACE in the caller's action tensor (:54), as upstream; reward (N,A,1), agent 0 only:
+5 past the door line, -1 when the opponent is past it, +1 when closer than at the previous step.  The constructor
hard-sets the scales 5 / 1 / 1 after copying the configured ones (:18-20).  Upstream's distance term broadcasts the target
over x AND y and only runs for num_envs in {1, 2}; the same arithmetic is applied here for any num_envs."""
import torch

from .empty_wrapper import FusedTaskWrapper


class Go1RotationWrapper(FusedTaskWrapper):
    task = "rotation"

    def __init__(self, env):
        super().__init__(env)
        self.success_reward_scale = 5
        self.distance_reward_scale = 1
        self.punishment_scale = 1

    def _obs_dim(self):
        return 12

    def step(self, action):
        action[:, 1, 1:] = -action[:, 1, 1:]
        obs, rew, done, info = super().step(action)
        return obs, rew.reshape(self.num_envs, self.num_agents, 1), done, info
