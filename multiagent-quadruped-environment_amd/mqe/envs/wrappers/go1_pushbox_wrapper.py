"""go1pushbox task wrapper (reference mqe/envs/wrappers/go1_pushbox_wrapper.py:10-88): obs (N,A,20+A) = [one-hot id,
own pos+rpy, other's pos+rpy, gate xy, box xy rel. env origin, box quaternion]; reward = box x-displacement since the
previous step (0 on the first step and for envs that reset), broadcast to the agents.  The reference constructor sets
`box_x_movement_reward_scale = 1` after copying the configured 10 (:20), so the effective scale is 1."""
from .empty_wrapper import FusedTaskWrapper


class Go1PushboxWrapper(FusedTaskWrapper):
    task = "pushbox"

    def __init__(self, env):
        super().__init__(env)
        self.box_x_movement_reward_scale = 1

    def _obs_dim(self):
        return 20 + self.num_agents
