"""go1sheep-easy/-hard task wrapper (reference mqe/envs/wrappers/go1_sheep_wrapper.py:8-118):
obs (N,A,14+2P+A) = [id, own pos+rpy, other's pos+rpy, gate xy, all sheep xy]."""
from .empty_wrapper import FusedTaskWrapper


class Go1SheepWrapper(FusedTaskWrapper):
    task = "sheep"

    def _obs_dim(self):
        return 14 + 2 * self.env.cfg.env.num_npcs + self.num_agents
