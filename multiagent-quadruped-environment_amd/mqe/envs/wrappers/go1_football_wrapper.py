"""go1football-defender task wrapper (reference mqe/envs/wrappers/go1_football_wrapper.py:8-91): only agents 0,1
are exposed; obs (N,2,20) = [id2, own pos+rpy, other's, ball pos rel. env origin, ball vel]."""
from .empty_wrapper import FusedTaskWrapper


class Go1FootballDefenderWrapper(FusedTaskWrapper):
    task = "football_defender"
    wrapper_agents = 2

    def _obs_dim(self):
        return 18 + self.num_agents
