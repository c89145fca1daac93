"""go1tug task wrapper (reference mqe/envs/wrappers/go1_tug_wrapper.py:9-136).  obs (N,A,10) = [own pos+rpy, slider (pos, vel),
distance to the slider's rim centre line, slider pos again], agent 1's copy mirrored (entries 1, 4, 6, 9 negated, the slider
velocity is not); agent 1's y / yaw commands negated IN PLACE and -- unlike every other wrapper -- the actions are NOT clipped
before the [2, .5, .5] scale (:60,70); reward (N,A,1), agent 0 only: slider on the opponent's side (+), on its own side (-),
approaching the slider (+) or not (- 2^distance).  After an env reset the wrapper re-zeroes the slider for two more steps
(`reset_dic`, :61-71); that and six running position logs in `reward_buffer` happen inside the fused step."""
from .empty_wrapper import FusedTaskWrapper


class Go1TugWrapper(FusedTaskWrapper):
    task = "tug"

    def _obs_dim(self):
        return 10

    def step(self, action):
        action[:, 1, 1:] = -action[:, 1, 1:]
        obs, rew, done, info = super().step(action)
        return obs, rew.reshape(self.num_envs, self.num_agents, 1), done, info
