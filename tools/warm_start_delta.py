#!/usr/bin/env python3
"""What would warm-started contact impulses change?  (CPU only: the experiment lives in the oracle -- MQO_WARM_START=<factor> at creation:
every contact of a substep starts its sweep from factor x the impulse the matching contact of the previous substep ended with, matched by
(kind, actors, links) and position within 2 cm; the specification and the HIP engine start every sweep from zero.)

  * free: go1gate, same seeded resets and random wrapper actions, a cold and a warm f32 oracle side by side: base-position difference after
    5 .. STEPS steps, falls, mean base height, mean vertical foot force -- the same aggregate measures as tests/solver_delta.py, so the
    number can be read against profiles/r04_solver_delta.json (what the choice between two legitimate solvers moves);
  * known answers (tests/contact_rich.py on the f64 oracle, cold / warm): the trot's time-step convergence and impulse balance, the drop
    onto four feet, the box on the ramp at the friction limit.
Usage: python tools/warm_start_delta.py [out.json] [N = 256] [steps = 200] [factor = 1.0]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import numpy as np
import torch
import contact_rich as cr
from helpers import make_desc, oracle_engine
from mqe.engine import abi

OUT = sys.argv[1] if len(sys.argv) > 1 else None
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
FACTOR = sys.argv[4] if len(sys.argv) > 4 else "1.0"


def engine(warm, f64=False):
    def make(d, k):
        if warm:
            os.environ["MQO_WARM_START"] = FACTOR
        try:
            return oracle_engine(d, k, f64=f64)
        finally:
            os.environ.pop("MQO_WARM_START", None)
    return make


def q(x):
    x = torch.as_tensor(x).flatten().float()
    return {"median": float(x.median()), "p90": float(x.quantile(0.9)), "p99": float(x.quantile(0.99)), "max": float(x.max())}


out = {"factor": float(FACTOR), "free": {}, "known_answers": {}}
for solver, st in (("tgs", 1), ("pgs", 0)):
    eng = {}
    for name in ("cold", "warm"):
        d, k, _ = make_desc("go1gate", N, solver_type=st)
        eng[name] = engine(name == "warm")(d, k)
        eng[name].reset_all()
    A = d.num_agents
    g = torch.Generator().manual_seed(7)
    rec = {"envs": N, "resets": {"cold": 0, "warm": 0}, "reset_flag_mismatches": 0, "pos_dev_m": {}}
    hs = {"cold": 0.0, "warm": 0.0}
    fs = {"cold": 0.0, "warm": 0.0}
    for t in range(1, STEPS + 1):
        a = torch.rand(N, A, 3, generator=g) * 2 - 1
        for nm in eng:
            eng[nm].step(a)
            rec["resets"][nm] += int(eng[nm].tensor(abi.T_RESET_BUF).sum())
            hs[nm] += float(eng[nm].tensor(abi.T_ROOT_STATE)[:, :A, 2].mean())
            cf = eng[nm].tensor(abi.T_CONTACT_FORCE).reshape(N, A, abi.NREP, 3)
            fs[nm] += float(cf[:, :, 1:, 2].sum(dim=2).mean())          # vertical force on the four feet of a robot
        rec["reset_flag_mismatches"] += int((eng["cold"].tensor(abi.T_RESET_BUF) != eng["warm"].tensor(abi.T_RESET_BUF)).sum())
        if t in (1, 5, 20, 50, 100, 200, 400):
            dev = (eng["cold"].tensor(abi.T_ROOT_STATE)[:, :A, :3] - eng["warm"].tensor(abi.T_ROOT_STATE)[:, :A, :3]).abs().amax(dim=(1, 2))
            rec["pos_dev_m"][str(t)] = q(dev)
    rec["mean_base_height_m"] = {nm: hs[nm] / STEPS for nm in hs}
    rec["mean_vertical_foot_force_N"] = {nm: fs[nm] / STEPS for nm in fs}
    out["free"][solver] = rec
    print(solver, json.dumps(rec), flush=True)
    for e in eng.values():
        e.close()

for name in ("cold", "warm"):
    mk = engine(name == "warm", f64=True)
    ka = {}
    ref = cr.trot(engine(False, f64=True), 0.000625)
    for dt in (0.005, 0.0025):
        r = cr.trot(mk, dt)
        ka[f"trot_dt_{dt}"] = {"median_dist_to_finest_cold_m": float(np.median(np.linalg.norm(r["pos"] - ref["pos"], axis=-1))),
                               "impulse_balance_rel": float(np.abs(r["impulse"] / r["impulse_expected"] - 1.0).max()), "fell": int(r["fell"].sum())}
    r = cr.drop(mk)
    ka["drop"] = {"fz_end_over_weight": float(np.mean(r["fz_end"] / r["weight"])), "ke_tail_over_fall": float(r["ke"][-40:].max() / r["ke_fall"]), "rebound_m": float(r["rebound"]),
                  "z_end_m": float(np.abs(r["z_end"]).max())}
    lim = cr.friction_frame_limit(0.5, False)
    st_, sl_ = cr.box_on_ramp(mk, 0.93 * lim, False, 0.5), cr.box_on_ramp(mk, 1.1 * lim, False, 0.5)
    ka["ramp"] = {"stick_slid_m": float(st_["slid"]), "slide_speed_mps": float(sl_["speed"]), "slide_gap_m": float(sl_["gap"])}
    out["known_answers"][name] = ka
    print(name, json.dumps(ka), flush=True)
if OUT:
    json.dump(out, open(OUT, "w"), indent=1)
