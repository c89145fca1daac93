"""Per-phase shader-clock timestamps of k_simulate for one env (debug launch, one wavefront on an idle GPU)."""
import ctypes as C
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import torch
from helpers import make_desc, hip_engine
from mqe.engine import abi
task = sys.argv[1] if len(sys.argv) > 1 else "go1gate"
d, k, _ = make_desc(task, 64)
e = hip_engine(d, k)
e.reset_all()
Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
for t in range(40):
    e.step(torch.zeros(64, Aw, 3, device="cuda"))
torch.cuda.synchronize()
names = ["load", "FK", "inertia+shuffles+Mcols", "leg blocks", "schur 6x6", "factor rows", "v*", "spheres", "one-sided contacts",
         "two-actor contacts", "side records", "-", "sweep", "dv = T w, limits", "end"]
for rep in range(2):
    minv, con = e.debug_dynamics(3, 0)
    t = (C.c_longlong * 16)()
    e.lib.mqe_debug_times(t)
    t = list(t)
    print("contacts:", len(con), "total", (t[14] - t[0]) * 0.01, "us (100 MHz wall clock, 10 ns per tick below)")
    for i in range(14):
        print(f"  {names[i]:28s} {t[i + 1] - t[i]:8d}")
