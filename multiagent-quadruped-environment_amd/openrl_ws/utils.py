"""OpenRL adapter (reference openrl_ws/utils.py:31-155,230-264): numpy in / numpy out around the task wrapper.
`openrl` itself is a third-party trainer above the hot path; nothing here imports it, so the adapter also serves any
other numpy-based trainer.  `step_torch` is the device-resident variant (no PCIe round trip)."""
import argparse
from typing import Any, Dict, Optional

import numpy as np
import torch

from mqe.envs.utils import make_mqe_env
from mqe.envs.wrappers.spaces import Wrapper
from mqe.utils.helpers import finish_args


def make_env(args, custom_cfg=None, single_agent=False):
    env, env_cfg = make_mqe_env(args.task, args, custom_cfg=custom_cfg)
    if single_agent:
        env = SingleAgentWrapper(env)
    return mqe_openrl_wrapper(env), env_cfg


class mqe_openrl_wrapper(Wrapper):
    def __init__(self, env):
        self.env = env
        self.agent_num = env.num_agents
        self.parallel_env_num = env.num_envs
        self.action_space = env.action_space
        self.observation_space = env.observation_space

    def reset(self, **kwargs):
        return self.env.reset().cpu().numpy()

    def step(self, actions, extra_data: Optional[Dict[str, Any]] = None):
        dev = getattr(self.env, "device", "cpu")
        a = torch.from_numpy(0.5 * actions).to(dev).clip(-1, 1).float()
        obs, reward, termination, info = self.env.step(a)
        obs = obs.cpu().numpy()
        rewards = reward.cpu().unsqueeze(-1).numpy()
        dones = termination.cpu().unsqueeze(-1).repeat(1, self.agent_num).numpy().astype(bool)
        return obs, rewards, dones, [{} for _ in range(dones.shape[0])]

    def step_torch(self, actions: torch.Tensor):
        """Same transform, tensors stay on the device: (obs (N,A,D), reward (N,A,1), done (N,A) bool)."""
        obs, reward, termination, info = self.env.step((0.5 * actions).clip(-1, 1))
        return obs, reward.unsqueeze(-1), termination.unsqueeze(-1).repeat(1, self.agent_num)

    def close(self, **kwargs):
        return self.env.close()

    @property
    def use_monitor(self):
        return False

    def batch_rewards(self, buffer):
        rb = self.env.reward_buffer
        step_count = rb["step count"]
        out = {"average step reward": 0}
        for k in list(rb.keys()):
            if k == "step count":
                continue
            out[k] = rb[k] / (self.num_envs * step_count)
            if hasattr(self.env, "single_agent_reward_scale"):
                out[k] *= self.env.single_agent_reward_scale
            if "reward" in k or "punishment" in k:
                out["average step reward"] += out[k]
            rb[k] = 0
        rb["step count"] = 0
        return out


class MATWrapper(Wrapper):
    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, actions, extra_data: Optional[Dict[str, Any]] = None):
        return self.env.step(actions, extra_data)


class SingleAgentWrapper(Wrapper):
    """Presents N envs x A agents as N*A single-agent envs (reference openrl_ws/utils.py:127-155)."""

    def __init__(self, env):
        super().__init__(env)
        self.num_envs = env.num_envs * env.num_agents
        self.num_agents = 1
        self.single_agent_reward_scale = env.num_agents

    def reset(self, **kwargs):
        return self.env.reset(**kwargs).reshape(self.num_envs, 1, -1)

    def step(self, actions, extra_data=None):
        a = actions.reshape(self.env.num_envs, self.env.num_agents, -1)
        obs, reward, termination, info = self.env.step(a)
        term = termination.unsqueeze(1).repeat(1, self.env.num_agents).reshape(self.num_envs)
        return obs.reshape(self.num_envs, 1, -1), reward.reshape(self.num_envs, 1), term, info


def get_args(argv=None):
    """CLI of the reference's train/test scripts (openrl_ws/utils.py:230-264) minus OpenRL's own parser."""
    p = argparse.ArgumentParser()
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--task", type=str, default="go1gate")
    p.add_argument("--algo", type=str, default="ppo")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=str)
    p.add_argument("--headless", action="store_true", default=True)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--max_iterations", type=int)
    p.add_argument("--train_timesteps", type=int)
    p.add_argument("--use_wandb", action="store_true", default=False)
    p.add_argument("--use_tensorboard", action="store_true", default=False)
    p.add_argument("--exp_name", type=str, default="default")
    p.add_argument("--record_video", action="store_true", default=False)
    return finish_args(p.parse_args(argv))
