import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd"), os.path.join(ROOT, "oracle")]
import torch, numpy as np
from mqe.engine import abi
from helpers import make_desc, hip_engine, oracle_engine
N = 16
d1, k1, _ = make_desc("go1gate", N); d2, k2, _ = make_desc("go1gate", N)
eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
eh.reset_all(); eo.reset_all()
g = torch.Generator().manual_seed(17)
f = eo.lib.mqo_policy_forward
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for t in range(34):
    a = torch.rand(N, 2, 3, generator=g) * 2 - 1
    eh.step(a.cuda().contiguous()); eo.step(a)
    torch.cuda.synchronize()
    Hh, Ho = eh.history().cpu(), eo.history()
    dH = (Hh - Ho).abs()
    ah, ao = eh.tensor(abi.T_LAST_LOCO_ACTION).cpu(), eo.tensor(abi.T_LAST_LOCO_ACTION)
    # oracle forward on HIP's history
    ref = np.zeros((2 * N, 12), np.float32); lat = np.zeros(2, np.float32)
    Hn = np.ascontiguousarray(Hh.numpy())
    for i in range(2 * N):
        f(eo.h, Hn[i].ctypes.data, lat.ctypes.data, ref[i].ctypes.data)
    print(t, "hist dev", float(dH.max()), "frame of max", int(dH.amax(0).argmax()) // 70, "| hip vs oracle act", float((ah - ao).abs().max()),
          "| hip vs oracle-forward(hip history)", float((ah - torch.from_numpy(ref)).abs().max()), "| nonzero frames", int((Ho.view(-1, 30, 70).abs().amax((0, 2)) > 0).sum()))
    for k in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_LAST_LOCO_ACTION, abi.T_LAST_TWO_LOCO_ACTION, abi.T_ACTIONS, abi.T_ACT_HIST, abi.T_OBS_BAG, abi.T_GAIT_INDICES, abi.T_CLOCK_INPUTS, abi.T_LAST_ACTIONS):
        eh.tensor(k).copy_(eo.tensor(k).cuda())
