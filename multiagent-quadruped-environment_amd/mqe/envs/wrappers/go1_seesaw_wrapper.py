"""go1seesaw task wrapper (reference mqe/envs/wrappers/go1_seesaw_wrapper.py:8-120): obs (N,A,12+A)."""
from .empty_wrapper import FusedTaskWrapper


class Go1SeesawWrapper(FusedTaskWrapper):
    task = "seesaw"

    def _obs_dim(self):
        return 12 + self.num_agents
