"""mqe -- host-side mirror of MQE's Python plugin surface, backed by the MI355X-native HIP engine.

Import paths follow the reference (mqe.envs.utils.make_mqe_env, mqe.envs.configs.*, mqe.utils.helpers ...), so
`openrl_ws/train.py`-style callers keep working; nothing here depends on Isaac Gym or gym.
"""
import os

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
LEGGED_GYM_ENVS_DIR = os.path.join(LEGGED_GYM_ROOT_DIR, "mqe", "envs")
