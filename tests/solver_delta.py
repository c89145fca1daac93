#!/usr/bin/env python3
"""Row H's specification uncertainty: the two contact solvers of the engine (include/mqe_hip.h solver_type: 0 = velocity-level projected
Gauss-Seidel with the erp bias, the scheme of rounds 1-3; 1 = temporal Gauss-Seidel as PhysX publishes it, what sim.physx.solver_type = 1
asks for) on every task, on the HIP engine, same seeded resets and random actions.

  * one_step: every 10th step the temporal engine's whole state (mqe_state_save) is loaded into a velocity-level engine and both make the
    same env.step() (4 substeps = 20 ms): difference of base position / height / velocity and joint angles after that ONE step from
    identical walking / falling states -- what the choice of solver changes per policy step;
  * free: a velocity-level engine started from the same reset runs freely beside the temporal one: base-position difference after
    5 .. STEPS steps (contact dynamics amplify any difference: this measures how fast two legitimate solvers part, not an error),
    plus the aggregate behaviour of each -- falls (resets), mean base height, mean vertical foot force.
Usage (GPU box): python tests/solver_delta.py [N = 1024] [out.json] [steps = 200]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import torch
from helpers import make_desc, hip_engine
from mqe.engine import abi
from mqe.envs.utils import ENV_DICT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
MARKS = [m for m in (1, 5, 20, 50, 100, 200, 400) if m <= STEPS]


def q(x):
    x = x.flatten().float()
    return {"median": float(x.median()), "p90": float(x.quantile(0.9)), "p99": float(x.quantile(0.99)), "max": float(x.max())}


out = {}
for task in ENV_DICT:
    n = N if "sheep-hard" not in task else max(N // 4, 8)
    eng = {}
    for name, st in (("tgs", 1), ("pgs", 0), ("pgs_sync", 0)):
        d, k, _ = make_desc(task, n, solver_type=st)
        eng[name] = hip_engine(d, k)
        eng[name].reset_all()
    A = d.num_agents
    Aw = eng["tgs"].tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator().manual_seed(7)
    rec = {"envs": n, "agents": A, "free": {}, "resets": {"tgs": 0, "pgs": 0}, "reset_flag_mismatches": 0}
    one = {"base_pos_m": [], "base_height_m": [], "base_linvel_mps": [], "base_angvel_radps": [], "joint_pos_rad": [], "joint_vel_radps": []}
    hsum = {"tgs": 0.0, "pgs": 0.0}
    fsum = {"tgs": 0.0, "pgs": 0.0}
    t0 = time.time()
    for t in range(1, STEPS + 1):
        a = (torch.rand(n, Aw, 3, generator=g) * 2 - 1).cuda().contiguous()
        sync = t % 10 == 0
        if sync:
            eng["pgs_sync"].load_state(eng["tgs"].save_state())
        eng["tgs"].step(a); eng["pgs"].step(a)
        if sync:
            eng["pgs_sync"].step(a)
        torch.cuda.synchronize()
        rt, rp = eng["tgs"].tensor(abi.T_ROOT_STATE)[:, :A], eng["pgs"].tensor(abi.T_ROOT_STATE)[:, :A]
        rec["reset_flag_mismatches"] += int((eng["tgs"].tensor(abi.T_RESET_BUF) != eng["pgs"].tensor(abi.T_RESET_BUF)).sum())
        for nm in ("tgs", "pgs"):
            rec["resets"][nm] += int(eng[nm].tensor(abi.T_RESET_BUF).sum())
            hsum[nm] += float(eng[nm].tensor(abi.T_ROOT_STATE)[:, :A, 2].mean())
            cf = eng[nm].tensor(abi.T_CONTACT_FORCE)[:, :17 * A].reshape(n, A, 17, 3)
            fsum[nm] += float(cf[:, :, [4, 8, 12, 16], 2].sum(-1).mean())
        if sync:
            keep = ~(eng["tgs"].tensor(abi.T_RESET_BUF).bool() | eng["pgs_sync"].tensor(abi.T_RESET_BUF).bool())     # envs that were not reset in this step
            rs = eng["pgs_sync"].tensor(abi.T_ROOT_STATE)[:, :A]
            dt_, ds_ = eng["tgs"].tensor(abi.T_DOF_STATE)[:, :12 * A], eng["pgs_sync"].tensor(abi.T_DOF_STATE)[:, :12 * A]
            one["base_pos_m"].append((rt[..., :3] - rs[..., :3]).norm(dim=-1)[keep].flatten().cpu())
            one["base_height_m"].append((rt[..., 2] - rs[..., 2]).abs()[keep].flatten().cpu())
            one["base_linvel_mps"].append((rt[..., 7:10] - rs[..., 7:10]).norm(dim=-1)[keep].flatten().cpu())
            one["base_angvel_radps"].append((rt[..., 10:13] - rs[..., 10:13]).norm(dim=-1)[keep].flatten().cpu())
            one["joint_pos_rad"].append((dt_[..., 0] - ds_[..., 0]).abs()[keep].flatten().cpu())
            one["joint_vel_radps"].append((dt_[..., 1] - ds_[..., 1]).abs()[keep].flatten().cpu())
        if t in MARKS:
            dev = (rt[..., :3] - rp[..., :3]).norm(dim=-1).amax(dim=1)
            rec["free"][f"base_pos_diff_m_step{t}"] = q(dev.cpu())
    rec["one_step_from_identical_states"] = {k: q(torch.cat(v)) for k, v in one.items()}
    rec["mean_base_height_m"] = {k: round(v / STEPS, 5) for k, v in hsum.items()}
    rec["mean_vertical_foot_force_N_per_robot"] = {k: round(v / STEPS, 3) for k, v in fsum.items()}
    rec["contact_list_overflows"] = {nm: int(eng[nm].tensor(abi.T_CONTACT_OVERFLOW).sum()) for nm in ("tgs", "pgs")}
    rec["seconds"] = round(time.time() - t0, 1)
    out[task] = rec
    print(task, json.dumps(rec), flush=True)
    del eng
if len(sys.argv) > 2:
    json.dump({"what": __doc__, "steps": STEPS, "tasks": out}, open(sys.argv[2], "w"), indent=1)
