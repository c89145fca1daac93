"""Boundary plumbing: class_to_dict / merge_dict / set_seed / make_env
(reference mqe/utils/helpers.py:46-61, :81-91, :237-261).  No Isaac Gym: `sim_params` is a plain namespace."""
import os
import random
import types

import numpy as np
import torch


def class_to_dict(obj):
    if not hasattr(obj, "__dict__") or isinstance(obj, dict):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def merge_dict(this: dict, other: dict):
    merged = this.copy()
    merged.update(other)
    return merged


_ENGINE_SEED = 0


def engine_seed():
    """Seed of the engine's counter RNG (reset noise, domain parameters, pushes, the sheep's random walk): the value of the last
    set_seed() call -- the reference's draws follow torch's global generator, which the same call seeds (helpers.py:81-91).  An
    env-sharded run must use one seed on all ranks; draws are keyed by the GLOBAL env id, so the shards then agree."""
    return _ENGINE_SEED


def set_seed(seed):
    global _ENGINE_SEED
    if seed == -1:
        seed = np.random.randint(0, 10000)
    _ENGINE_SEED = int(seed)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    return seed


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None and getattr(args, "num_envs", None) is not None:
        env_cfg.env.num_envs = args.num_envs
    return env_cfg, cfg_train


def parse_sim_params(args, cfg):
    """cfg["sim"] (a dict) -> attribute namespace; the PhysX-specific switches of the reference
    (use_gpu, subscenes, num_threads; helpers.py:93-115) are accepted and recorded, nothing consumes them."""
    def ns(d):
        return types.SimpleNamespace(**{k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    sp = ns(cfg.get("sim", {}))
    sp.use_gpu_pipeline = getattr(args, "use_gpu_pipeline", True)
    if hasattr(sp, "physx"):
        sp.physx.use_gpu = getattr(args, "use_gpu", True)
        sp.physx.num_subscenes = getattr(args, "subscenes", 0)
        if getattr(args, "num_threads", 0) > 0:
            sp.physx.num_threads = args.num_threads
    return sp


def get_args(argv=None):
    """Same flags as the reference CLI (helpers.py:168-194 + gymutil.parse_arguments)."""
    import argparse
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--task", type=str, default="go1gate")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--max_iterations", type=int)
    p.add_argument("--record_video", action="store_true", default=False)
    args = p.parse_args(argv)
    finish_args(args)
    return args


def finish_args(args):
    """Derived fields the reference computes after parsing (helpers.py:189-193, gymutil)."""
    dev = args.sim_device.lower()
    args.sim_device_type, args.compute_device_id = (dev.split(":")[0], int(dev.split(":")[1])) if ":" in dev else (dev, 0)
    args.use_gpu_pipeline = args.pipeline.lower() in ("gpu", "cuda")
    args.physics_engine = 1  # SIM_PHYSX in the reference; here: the HIP engine
    args.use_gpu = args.sim_device_type == "cuda"
    args.sim_device_id = args.compute_device_id
    args.sim_device = args.sim_device_type + (f":{args.sim_device_id}" if args.sim_device_type == "cuda" else "")
    return args


def make_env(task_class, env_cfg, args=None):
    if args is None:
        args = get_args()
    env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
    set_seed(args.seed)          # also the engine's seed: see engine_seed()
    sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
    env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                     sim_device=args.sim_device, headless=args.headless)
    return env, env_cfg
