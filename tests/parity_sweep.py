#!/usr/bin/env python3
"""HIP engine vs CPU oracle on every task at a few hundred envs: fused steps from the seeded reset distribution with the same
random actions; prints / writes per task the deviation of the robots' base positions (median and 99th percentile over envs)
after 5, 20 and 50 steps, the reset-flag mismatches and the largest policy-action difference at step 0.  Contact dynamics
amplify rounding differences, so the late numbers measure trajectory divergence, not arithmetic error (tests/ pin the
arithmetic on single steps).  Usage (GPU box): python tests/parity_sweep.py [N] [out.json] [steps = 50]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import torch
from helpers import make_desc, hip_engine, oracle_engine
from mqe.engine import abi
from mqe.envs.utils import ENV_DICT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 50
MARKS = [m for m in (5, 20, 50, 100, 200, 400) if m <= STEPS]
out = {}
for task in ENV_DICT:
    n = N if "sheep-hard" not in task else max(N // 4, 8)
    d1, k1, _ = make_desc(task, n)
    d2, k2, _ = make_desc(task, n)
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    A = d1.num_agents
    Aw = eo.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator().manual_seed(7)
    rec = {"envs": n, "agents": A}
    mism = 0
    t0 = time.time()
    for t in range(1, STEPS + 1):
        a = torch.rand(n, Aw, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        mism += int((eh.tensor(abi.T_RESET_BUF).cpu() != eo.tensor(abi.T_RESET_BUF)).sum())
        if t == 1:
            rec["max_policy_action_diff_step0"] = float((eh.tensor(abi.T_ACTIONS).cpu() - eo.tensor(abi.T_ACTIONS)).abs().max())
        if t in MARKS:
            dev = (eh.tensor(abi.T_ROOT_STATE).cpu()[:, :A, :3] - eo.tensor(abi.T_ROOT_STATE)[:, :A, :3]).abs().amax(dim=(1, 2))
            rec[f"pos_dev_m_step{t}"] = {"median": float(dev.median()), "p99": float(dev.quantile(0.99)), "max": float(dev.max()), "finite": bool(torch.isfinite(dev).all())}
    rec["reset_flag_mismatches_in_%d_steps" % STEPS] = mism
    rec["contact_list_overflows_hip_oracle"] = [int(eh.tensor(abi.T_CONTACT_OVERFLOW).sum()), int(eo.tensor(abi.T_CONTACT_OVERFLOW).sum())]
    rec["seconds"] = round(time.time() - t0, 1)
    out[task] = rec
    print(task, json.dumps(rec))
out["_config"] = {"contact_solver": os.environ.get("MQE_SOLVER", "tgs (default)"), "edge_contacts": os.environ.get("MQE_EDGE_CONTACTS", "3 (default)"),
                  "collision_model": os.environ.get("MQE_COLLISION_MODEL", "capsule (default)")}
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
