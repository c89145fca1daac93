"""Where the epilogue of k_substeps (the fused post-physics step) spends its time: MQE_PHASE_TIMES=1 taps inside post_body.
    python tools/dev/epilogue_taps.py [task = go1gate] [num_envs = 4096] [steps = 60]"""
import os, sys, ctypes as C
os.environ["MQE_PHASE_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import numpy as np, torch
from helpers import make_desc, hip_engine
from mqe.engine import abi
TASK = sys.argv[1] if len(sys.argv) > 1 else "go1gate"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
NAMES = ["act_hist store (end of substep 4 .. tap 0)", "fence, actions -> LDS", "loads + frame quantities", "flag / frame stores", "NPC rows -> LDS", "NPC script (sheep)",
         "reset", "observation rows -> LDS", "wrapper (lead lane)", "row flush", "history zero (resets)", "to the exit stamp"]


def run(n):
    d, k, _ = make_desc(TASK, n)
    e = hip_engine(d, k)
    e.reset_all()
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    ph, ep, wt = np.zeros((n, 4, 16), np.int64), np.zeros((n, 16), np.int64), np.zeros((n, 4), np.int64)
    acc, cnt = np.zeros(12), 0
    for t in range(steps):
        e.step(torch.rand(n, Aw, 3, device="cuda", generator=g) * 2 - 1)
        if t >= steps // 2:
            e._call("debug_phase_times", C.c_void_p(ph.ctypes.data))
            e._call("debug_epilogue_times", C.c_void_p(ep.ctypes.data))
            e._call("debug_wave_times", C.c_void_p(wt.ctypes.data))
            seq = np.concatenate([ph[:, 3, 15:16], ep[:, :11], wt[:, 1:2]], axis=1).astype(np.float64) * 0.01
            acc += np.diff(seq, axis=1).mean(axis=0); cnt += 1
    return acc / cnt


full, lone = run(N), run(256)
print(f"{TASK}: epilogue of k_substeps, mean over wavefronts and the last {steps - steps // 2} steps [us]")
print(f"{'section':48s} {'N=' + str(N):>10s} {'N=256':>10s}")
for i, nm in enumerate(NAMES):
    print(f"{nm:48s} {full[i]:10.2f} {lone[i]:10.2f}")
print(f"{'sum':48s} {full.sum():10.2f} {lone.sum():10.2f}")
