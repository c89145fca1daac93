"""Base task wrapper (reference mqe/envs/wrappers/empty_wrapper.py:4-18) + the device-resident reward log."""
import torch

from mqe.engine import abi
from mqe.engine.desc import REWARD_TERMS
from .spaces import Wrapper, Box  # noqa: F401


class RewardBuffer(dict):
    """`reward_buffer` of the reference wrappers: per-term reward sums + "step count".  The reference adds
    `torch.sum(term).cpu()` every step (a host sync per term, e.g. go1_sheep_wrapper.py:77,83,93,105,112); here the
    sums live on the device ([N, terms], accumulated in-kernel) and are reduced only when somebody reads a value
    (mqe_openrl_wrapper.batch_rewards, openrl_ws/utils.py:76-90).  Assigning 0 to a key clears its column."""

    def __init__(self, names, sums):
        super().__init__()
        self._col = {n: i for i, n in enumerate(names) if n is not None}
        self._sums = sums
        for n in names:
            if n is not None:
                dict.__setitem__(self, n, 0)
        dict.__setitem__(self, "step count", 0)

    def __getitem__(self, k):
        if k in self._col:
            return self._sums[:, self._col[k]].sum()
        return dict.__getitem__(self, k)

    def __setitem__(self, k, v):
        if k in self._col:
            self._sums[:, self._col[k]] = v
        else:
            dict.__setitem__(self, k, v)

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class EmptyWrapper(Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self.num_envs = env.num_envs
        self.num_agents = env.num_agents
        if hasattr(env.cfg.terrain, "BarrierTrack_kwargs"):
            self.BarrierTrack_kwargs = env.cfg.terrain.BarrierTrack_kwargs
        for key in dir(env.cfg.rewards.scales):
            if key[0] != "_" and "scale" in key:
                setattr(self, key, getattr(env.cfg.rewards.scales, key))
        self.obs_ids = torch.eye(self.num_agents, dtype=torch.float32, device=env.device).repeat(self.num_envs, 1).reshape(self.num_envs, self.num_agents, -1)


class FusedTaskWrapper(EmptyWrapper):
    """Shared by the four task wrappers: observation / reward come out of the engine's fused step."""
    task = "plain"
    obs_dim = 0
    wrapper_agents = None

    def __init__(self, env):
        super().__init__(env)
        if self.wrapper_agents is not None:
            self.num_agents = self.wrapper_agents
            self.obs_ids = torch.eye(self.num_agents, dtype=torch.float32, device=env.device).repeat(self.num_envs, 1).reshape(self.num_envs, self.num_agents, -1)
        assert env.task == self.task, f"{type(self).__name__} wraps task '{self.task}', env was built for '{env.task}'"
        dsc = env.engine.desc
        if dsc.num_command_dims != 3 or [dsc.command_src[c] for c in range(18)] != [-1] * 3 + [0, 1, 2] + [-1] * 12:
            # upstream's task wrappers multiply the (N, A, 3) action by a (1, 1, 3) scale (go1_*_wrapper.py step()): there is no room for the
            # action columns that command.cfg.{body_height, gait_freq, ...} add (go1.py:64-93); such a config runs behind EmptyWrapper / Go1.step
            raise NotImplementedError("command.cfg flags beyond the velocity command: the task wrappers carry exactly (x, y, yaw) per agent, "
                                      "as upstream; step the Go1 env itself (EmptyWrapper, e.g. go1plane)")
        self.observation_space = Box(low=-float("inf"), high=float("inf"), shape=(self._obs_dim(),), dtype=float)
        self.action_space = Box(low=-1, high=1, shape=(3,), dtype=float)
        self.action_scale = torch.tensor([[[2, 0.5, 0.5]]], device=env.device).repeat(self.num_envs, self.num_agents, 1)
        self._wobs = env.engine.tensor(abi.T_WRAPPER_OBS)
        self._wrew = env.engine.tensor(abi.T_WRAPPER_REWARD)
        self._wpack = env.engine.tensor(abi.T_WRAPPER_PACKED)        # obs | reward | done (N bytes) in one buffer
        assert self._wobs.shape[-1] == self.observation_space.shape[0]
        self.reward_buffer = RewardBuffer([n for _, n in REWARD_TERMS[self.task]], env.engine.tensor(abi.T_REWARD_SUMS))

    def _obs_dim(self):
        raise NotImplementedError

    def reset(self):
        self.env.reset()
        return self._wobs.clone()

    def step(self, action):
        # fresh tensors every step, like the reference: the HIP engine writes obs | reward | done of this step straight into a
        # new tensor (mqe_set_return_buffer; the engine's own MQE_T_WRAPPER_* buffer then keeps the previous contents); an
        # engine without that entry point is snapshotted with one copy
        eng = self.env.engine
        direct = hasattr(eng, "set_return_buffer")
        if direct:
            snap = torch.empty_like(self._wpack)
            eng.set_return_buffer(snap)
        try:
            self.env.step_fused(action.reshape(self.num_envs, self.num_agents, 3))
        finally:
            if direct:
                eng.set_return_buffer(None)
        dict.__setitem__(self.reward_buffer, "step count", dict.__getitem__(self.reward_buffer, "step count") + 1)
        if not direct:
            snap = self._wpack.clone()
        n, nr = self._wobs.numel(), self._wrew.numel()
        self.returned_batch = snap                                     # obs | reward | done (0/1): what a sharded runner all-gathers
        # all three returned tensors belong to this step alone (the reference builds a new reset_buf every step; env.reset_buf
        # is a live view of engine memory that the next step overwrites) and all three are VIEWS of the one buffer the step wrote:
        # the done flags are its byte tail, seen as torch.bool in place -- no torch kernel runs in a step
        done = snap[n + nr:].view(torch.uint8)[:self.num_envs].view(torch.bool)
        return snap[:n].view(self._wobs.shape), snap[n:n + nr].view(self._wrew.shape), done, self.env.extras
