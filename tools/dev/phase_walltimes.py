"""Where the time of a FULL k_substeps launch goes, phase by phase (MQE_PHASE_TIMES=1 -> mqe_debug_phase_times):
    python tools/dev/phase_walltimes.py [num_envs = 4096] [steps = 60] [task = go1gate | go1sheep-hard | go1football-defender]
Every wavefront of the fused decimation launch stamps the 100 MHz wall clock at the phase taps of each of its four substeps.  Printed:
the mean duration of each phase per substep [us] over all wavefronts, at the given batch (4096 = 4 wavefronts per SIMD, everything
resident) and, for comparison, with 256 envs (one wavefront per CU: the wavefront's own dependency chain)."""
import os, sys, ctypes as C
os.environ["MQE_PHASE_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import numpy as np, torch
from helpers import make_desc, hip_engine
from mqe.engine import abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
TASK = sys.argv[3] if len(sys.argv) > 3 else "go1gate"
NAMES = ["load", "FK", "inertia+Mcols", "(leg blocks)", "schur", "factor rows", "v*", "spheres/prims", "one-sided contacts", "two-actor / self",
         "side records", "(-)", "sweep", "dv, limits", "forces, integrate, store", "actuator net + logs (to the next substep)"]


def run(n):
    d, k, _ = make_desc(TASK, n)
    e = hip_engine(d, k)
    e.reset_all()
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    buf = np.zeros((n, 4, 16), np.int64)
    acc, cnt = np.zeros((4, 16)), 0
    tot, pro, epi, span = [], [], [], []
    wt = np.zeros((n, 4), np.int64)
    for t in range(steps):
        e.step(torch.rand(n, Aw, 3, device="cuda", generator=g) * 2 - 1)
        if t >= steps // 2:
            e._call("debug_phase_times", C.c_void_p(buf.ctypes.data))
            b = buf.astype(np.float64) * 0.01                       # us
            dur = np.zeros((n, 4, 16))
            dur[:, :, :15] = b[:, :, 1:16] - b[:, :, 0:15]           # phase i = tap i .. tap i + 1 ([15] = end of the substep)
            dur[:, :3, 15] = b[:, 1:, 0] - b[:, :3, 15]              # to the next substep's first tap: actuator network, logs
            acc += dur.mean(axis=0); cnt += 1
            tot.append((b[:, 3, 15] - b[:, 0, 0]).mean())
            e._call("debug_wave_times", C.c_void_p(wt.ctypes.data))      # entry / exit of every wavefront of the same launch
            w = wt.astype(np.float64) * 0.01
            pro.append((b[:, 0, 0] - w[:, 0]).mean()); epi.append((w[:, 1] - b[:, 3, 15]).mean()); span.append(w[:, 1].max() - w[:, 0].min())
    return acc / cnt, float(np.mean(tot)), (float(np.mean(pro)), float(np.mean(epi)), float(np.mean(span)))


full, tf, xf = run(N)
lone, tl, xl = run(256)
print(f"{TASK}, k_substeps with live phase taps; mean over wavefronts and over the last {steps - steps // 2} steps [us]")
print(f"{'phase':42s} {'N=' + str(N) + ' per substep':>18s} {'N=256 per substep':>18s}   ratio")
for i, nm in enumerate(NAMES):
    a, b = full[:, i].mean() if i < 15 else full[:3, i].mean(), lone[:, i].mean() if i < 15 else lone[:3, i].mean()
    print(f"{nm:42s} {a:18.2f} {b:18.2f}   {a / max(b, 1e-9):5.2f}")
print(f"{'first tap .. end of the 4th substep':42s} {tf:18.1f} {tl:18.1f}   {tf / tl:5.2f}")
print(f"{'kernel entry .. first tap (state load)':42s} {xf[0]:18.1f} {xl[0]:18.1f}")
print(f"{'end of the 4th substep .. exit (epilogue)':42s} {xf[1]:18.1f} {xl[1]:18.1f}")
print(f"{'first entry .. last exit of the launch':42s} {xf[2]:18.1f} {xl[2]:18.1f}")
print("per substep (all phases):", " ".join(f"{full[k].sum():.1f}" for k in range(4)), "|", " ".join(f"{lone[k].sum():.1f}" for k in range(4)))
