#!/usr/bin/env python3
"""Time-step convergence of row H on contact-rich motion (VERDICT r4 "Next" 5a): the float64 CPU specification (oracle/) runs a trotting
Go1 (tests/contact_rich.py::trot: PD control towards a continuous-time joint-target trajectory, 2 robots x N envs, 0.5 s from identical
states) at dt = 5 / 2.5 / 1.25 / 0.625 ms under both contact solvers, and -- when a GPU is present -- the HIP engine does the same.
Written to profiles/r05_dt_convergence.json: per step size the distance of base positions / joint angles to the finest run, mean base
height, fall count, the vertical contact impulse against the momentum theorem.  No GPU needed for the oracle part.
Usage: python tools/dt_convergence.py [out.json] [N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("multiagent-quadruped-environment_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import contact_rich as cr  # noqa: E402
from helpers import oracle_engine, hip_engine  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_dt_convergence.json")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
DTS = (0.005, 0.0025, 0.00125, 0.000625)
engines = {"oracle_f64": lambda d, k: oracle_engine(d, k, f64=True), "oracle_f32": lambda d, k: oracle_engine(d, k, f64=False)}
if torch.cuda.is_available():
    engines["hip"] = hip_engine
doc = {"_what": "tests/contact_rich.py::trot, go1gate, %d envs x 2 robots, 0.5 s, PD control (kp 20, kd 0.5) towards a continuous-time trot "
                "(0.15-0.4 rad, 1.5-3 Hz per robot); distances are to the float64 oracle's run at dt = 0.625 ms under the same solver" % N,
       "dt_ms": [1e3 * dt for dt in DTS], "solvers": {}}
for solver in ("tgs", "pgs"):
    os.environ["MQE_SOLVER"] = solver
    ref = None
    rows = {}
    for name, mk in engines.items():
        res = {dt: cr.trot(mk, dt, N=N) for dt in DTS[::-1]}
        if ref is None:
            ref = res[DTS[-1]]
        r = {}
        for dt in DTS:
            x = res[dt]
            dp = np.linalg.norm(x["pos"] - ref["pos"], axis=-1).ravel()
            dq = np.abs(x["q"] - ref["q"]).max(-1).ravel()
            r["%.3f ms" % (1e3 * dt)] = {"base_pos_m": {"median": float(np.median(dp)), "p90": float(np.percentile(dp, 90)), "max": float(dp.max())},
                                         "joint_rad": {"median": float(np.median(dq)), "max": float(dq.max())},
                                         "mean_base_height_m": float(x["height"].mean()), "falls": int(x["fell"].sum()),
                                         "contact_impulse_vs_momentum_theorem_rel": float(np.abs(x["impulse"] / x["impulse_expected"] - 1).max())}
        med = [r["%.3f ms" % (1e3 * dt)]["base_pos_m"]["median"] for dt in DTS]
        r["order_estimate"] = [float(np.log2(med[i] / med[i + 1])) for i in range(len(DTS) - 2) if med[i + 1] > 0]
        rows[name] = r
    doc["solvers"][solver] = rows
json.dump(doc, open(out, "w"), indent=1)
for s, rows in doc["solvers"].items():
    for name, r in rows.items():
        print(s, name, [("%.2e" % r["%.3f ms" % (1e3 * dt)]["base_pos_m"]["median"]) for dt in DTS], "order", ["%.2f" % o for o in r["order_estimate"]])
