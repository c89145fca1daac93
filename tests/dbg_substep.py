import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import torch, numpy as np
from test_gpu_parity import _pair, _randomize
from mqe.engine import abi
task = sys.argv[1]; N = int(sys.argv[2])
eh, eo, d = _pair(task, N)
for seed, drop in ((1, 0.0), (2, 0.11), (3, 0.2)):
    _randomize(eh, eo, seed, drop=drop)
    ncs = [len(eo.debug_dynamics(e, 0)[2]) for e in range(N)]
    eh.simulate(); eo.simulate(); torch.cuda.synchronize()
    dq = (eh.tensor(abi.T_DOF_STATE)[..., 0].cpu() - eo.tensor(abi.T_DOF_STATE)[..., 0]).abs().amax(dim=1)
    dr = (eh.tensor(abi.T_ROOT_STATE)[..., :3].cpu() - eo.tensor(abi.T_ROOT_STATE)[..., :3]).abs().amax(dim=(1, 2))
    bad = [(e, ncs[e], round(dq[e].item(), 5), round(dr[e].item(), 5)) for e in range(N) if dq[e] > 2e-5 or dr[e] > 2e-5]
    print("seed", seed, "drop", drop, "max nc", max(ncs), "bad envs (env, nc, dq, dr):", bad[:10])
    for e, nc, _, _ in bad[:2]:
        _, _, co = eo.debug_dynamics(e, 0)
        print("   contacts env", e, co[:, :4].astype(int).tolist())
