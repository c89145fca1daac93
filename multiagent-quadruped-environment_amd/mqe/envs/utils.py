"""Task registry and factory: the drop-in boundary of the package (same names and call signatures as the reference's
mqe/envs/utils.py:38-134: `ENV_DICT[task] = {"class", "config", "wrapper"}`, `make_mqe_env(task, args, custom_cfg)`,
`custom_cfg(args)`).  The registry itself is a table of names resolved on import."""
from importlib import import_module
from typing import Tuple

from mqe.envs.configs._build import cfg
from mqe.envs.field.legged_robot_field import LeggedRobotField
from mqe.envs.field.legged_robot_field_config import LeggedRobotFieldCfg
from mqe.utils import get_args, make_env  # noqa: F401

#        task                      environment class (module:name)                   config entry              wrapper (module:name)
_TASKS = (
    ("go1plane",             "go1.go1:Go1",                                    "Go1PlaneCfg",            "empty_wrapper:EmptyWrapper"),
    ("go1gate",              "go1.go1:Go1",                                    "Go1GateCfg",             "go1_gate_wrapper:Go1GateWrapper"),
    ("go1sheep-easy",        "npc.go1_sheep:Go1Sheep",                         "SingleSheepCfg",         "go1_sheep_wrapper:Go1SheepWrapper"),
    ("go1sheep-hard",        "npc.go1_sheep:Go1Sheep",                         "NineSheepCfg",           "go1_sheep_wrapper:Go1SheepWrapper"),
    ("go1football-defender", "npc.go1_football_defender:Go1FootballDefender",  "Go1FootballDefenderCfg", "go1_football_wrapper:Go1FootballDefenderWrapper"),
    ("go1football-1vs1",     "npc.go1_object:Go1Object",                       "Go1Football1vs1Cfg",     "go1_football_wrapper:Go1FootballGameWrapper"),
    ("go1football-2vs2",     "npc.go1_object:Go1Object",                       "Go1Football2vs2Cfg",     "go1_football_wrapper:Go1FootballGameWrapper"),
    ("go1seesaw",            "npc.go1_object:Go1Object",                       "Go1SeesawCfg",           "go1_seesaw_wrapper:Go1SeesawWrapper"),
    ("go1pushbox",           "npc.go1_object:Go1Object",                       "Go1PushboxCfg",          "go1_pushbox_wrapper:Go1PushboxWrapper"),
    ("go1revolvingdoor",     "npc.go1_object:Go1Object",                       "Go1RotationCfg",         "go1_rotation_wrapper:Go1RotationWrapper"),
    ("go1tug",               "npc.go1_object:Go1Object",                       "Go1TugCfg",              "go1_tug_wrapper:Go1TugWrapper"),
    ("go1bridge",            "npc.go1_object:Go1Object",                       "Go1BridgeCfg",           "go1_bridge_wrapper:Go1BridgeWrapper"),
    ("go1wrestling",         "npc.go1_object:Go1Object",                       "Go1WrestlingCfg",        "go1_wrestling_wrapper:Go1WrestlingWrapper"),
)


def _resolve(package, spec):
    module, name = spec.split(":")
    return getattr(import_module(f"{package}.{module}"), name)


ENV_DICT = {task: {"class": _resolve("mqe.envs", env), "config": cfg(entry), "wrapper": _resolve("mqe.envs.wrappers", wrapper)}
            for task, env, entry, wrapper in _TASKS}


def make_mqe_env(env_name: str, args=None, custom_cfg=None) -> Tuple[LeggedRobotField, LeggedRobotFieldCfg]:
    """environment of task `env_name` inside its task wrapper, and the config it was built from"""
    task = ENV_DICT[env_name]
    if callable(custom_cfg):
        task["config"] = custom_cfg(task["config"])
    env, env_cfg = make_env(task["class"], task["config"], args)
    return task["wrapper"](env), env_cfg


def custom_cfg(args):
    """config hook of the training scripts: --num_envs and --record_video override the task defaults"""
    def apply(cfg_cls: LeggedRobotFieldCfg):
        n = getattr(args, "num_envs", None)
        if n is not None:
            cfg_cls.env.num_envs = n
        cfg_cls.env.record_video = getattr(args, "record_video", False)
        return cfg_cls
    return apply
