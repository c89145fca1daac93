"""debug helper: one substep HIP vs oracle from the rough states of tests/test_gpu_parity.py::test_single_substep_matches_oracle; prints the worst env"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("multiagent-quadruped-environment_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from mqe.engine import abi
from test_gpu_parity import _pair, _randomize
task, N = sys.argv[1], int(sys.argv[2])
eh, eo, d = _pair(task, N)
for seed, drop in ((1, 0.0), (2, 0.11), (3, 0.2)):
    _randomize(eh, eo, seed, drop=drop)
    q0 = eo.tensor(abi.T_DOF_STATE).clone(); r0 = eo.tensor(abi.T_ROOT_STATE).clone()
    eh.simulate(); eo.simulate(); torch.cuda.synchronize()
    dq = (eh.tensor(abi.T_DOF_STATE)[..., 0].cpu() - eo.tensor(abi.T_DOF_STATE)[..., 0]).abs()
    e = int(dq.max(dim=1).values.argmax())
    print("seed", seed, "max dof pos err", float(dq.max()), "env", e, "n envs > 2e-5:", int((dq.max(dim=1).values > 2e-5).sum()))
    if dq.max() > 2e-5:
        A = d.num_agents
        print(" dof pos err per joint", (dq[e] * 1e5).round().tolist())
        print(" oracle q1", eo.tensor(abi.T_DOF_STATE)[e, :, 0].tolist())
        print(" oracle v1", eo.tensor(abi.T_DOF_STATE)[e, :, 1].tolist())
        print(" hip    v1", eh.tensor(abi.T_DOF_STATE)[e, :, 1].cpu().tolist())
        print(" q0", q0[e, :, 0].tolist()); print(" v0", q0[e, :, 1].tolist())
        print(" lower", [d.robot.dof_lower[j] for j in range(12)], "upper", [d.robot.dof_upper[j] for j in range(12)])
        eo.tensor(abi.T_DOF_STATE).copy_(q0); eo.tensor(abi.T_ROOT_STATE).copy_(r0)
        for r in range(A):
            _, _, co = eo.debug_dynamics(e, r)
        print(" contacts (actA, linkA, actB, linkB, sd):", [[int(c[0]), int(c[1]), int(c[2]), int(c[3]), round(float(c[4]), 5)] for c in co])
        print(" root pose err", (eh.tensor(abi.T_ROOT_STATE)[e, :, :7].cpu() - eo.tensor(abi.T_ROOT_STATE)[e, :, :7]).abs().max().item())
        cfh = eh.tensor(abi.T_CONTACT_FORCE)[e].cpu().reshape(-1, 3); cfo = eo.tensor(abi.T_CONTACT_FORCE)[e].reshape(-1, 3)
        for b in range(cfh.shape[0]):
            if cfo[b].abs().max() > 0 or cfh[b].abs().max() > 0:
                print("  body", b, "hip", [round(float(x), 2) for x in cfh[b]], "oracle", [round(float(x), 2) for x in cfo[b]])
        print(" root hip", eh.tensor(abi.T_ROOT_STATE)[e].cpu().tolist())
        print(" root ora", eo.tensor(abi.T_ROOT_STATE)[e].tolist())
