"""go1bridge task wrapper (reference mqe/envs/wrappers/go1_bridge_wrapper.py:8-80).  obs (N,A,12) = [own pos+rpy, other's
pos+rpy], agent 1's copy re-expressed as walking the bridge the other way (x -> |x0_reset + x1_reset| - x for both robots,
pitch negated); agent 1's y / yaw commands negated IN PLACE in the caller's tensor (:43); reward (N,A), agent 0 only:
+10 when the opponent is below 0.5 m, -1 when agent 0 is, +1 once agent 0 is past the opponent's start."""
from .empty_wrapper import FusedTaskWrapper


class Go1BridgeWrapper(FusedTaskWrapper):
    task = "bridge"

    def _obs_dim(self):
        return 12

    def step(self, action):
        action[:, 1, 1:] = -action[:, 1, 1:]
        return super().step(action)
