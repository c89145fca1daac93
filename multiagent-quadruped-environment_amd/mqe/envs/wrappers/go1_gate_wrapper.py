"""go1gate task wrapper.  NOTE: at the surveyed commit the reference's wrapper is stubbed (its reset()/step()
return obs = 0, reward = 0; mqe/envs/wrappers/go1_gate_wrapper.py:58-76,155-156).  We implement what its commented-out body
computes (:41-54,64-67,78-154): obs (N,A,14+A) = [one-hot id, own pos+rpy, other's pos+rpy, gate xy]; reward = progress to
target + 5*success - 2*contact - 0.025/d^2 (d^2<0.25), summed over agents and broadcast.  Pinned by reference vectors:
tools/gen_golden.py::gen_gate_wrapper activates that block in memory and records its outputs (tests/golden/wrapper_gate.npz)."""
from .empty_wrapper import FusedTaskWrapper


class Go1GateWrapper(FusedTaskWrapper):
    task = "gate"

    def _obs_dim(self):
        return 14 + self.num_agents
