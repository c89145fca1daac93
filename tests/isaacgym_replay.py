"""Replay of Isaac Gym captures (tools/capture_isaacgym_trace.py -> tests/golden/isaacgym_<task>.npz) through an engine of this
build: the only road from row H ("the physics substep") to the reference's own numbers.  Used by
tests/test_isaacgym_trace.py (CPU oracle) and tests/test_gpu_isaacgym_trace.py (HIP engine).

Two replays of a capture {root[k], dof[k], tau[k]}:
  * one-step:      state := recorded state k, torques := tau[k], ONE substep, compared with recorded state k+1 (solver error alone);
  * free-running:  state := recorded state 0, then the recorded torque sequence with nothing re-synchronised (trajectory error).

STATED TOLERANCES (what the tests assert; a capture that breaks them is a finding about the physics specification of
DESIGN.md section 4 -- solver type, ERP, friction model, collision primitives -- not about the replay):
"""
import glob
import json
import os

import numpy as np
import torch

from mqe.engine import abi
from mqe.utils import urdf_model
from helpers import GOLD, make_desc

# Derived (round 4) from what the engine's own two contact solvers -- velocity-level projected Gauss-Seidel and the temporal Gauss-Seidel
# that sim.physx.solver_type = 1 asks for -- differ by from IDENTICAL states (tests/solver_delta.py -> profiles/r04_solver_delta.json,
# 13 tasks x 1024 envs x 20 sampled policy steps; worst env per policy step of 4 substeps: base position 1.4-4.9 mm, base velocity
# 0.13-0.35 m/s, joint angle 0.02-0.036 rad, joint speed 6.5-7.4 rad/s; medians 6e-7 m, 2e-5 m/s, 2e-5 rad, 7e-4 rad/s).  A capture of
# the real PhysX may differ from either by about as much as they differ from each other, so a bound is 2 x that worst case, per substep
# (a quarter of the policy step's): the replay asserts the MAXIMUM over envs and substeps.
# one substep (5 ms) from the recorded state
TOL_ONE_STEP = dict(base_pos=2.5e-3,      # [m]     2 x 4.9 mm / 4
                    base_vel=0.18,        # [m/s]   2 x 0.35 / 4 (an impulse of 2.2 N s on the 12.7 kg robot)
                    joint_pos=1.8e-2,     # [rad]   2 x 0.036 / 4
                    joint_vel=3.7)        # [rad/s] 2 x 7.4 / 4 (contact onset moves a leg's joints by this much in either solver)
# free-running for the whole capture (default 40 policy steps = 0.8 s): the robots must stay on their feet the same way.  The two solvers
# are 1.1-1.4 cm (99th percentile) / 2.3-3.0 cm (worst env) apart after 50 free steps: 2 x the worst env
TOL_FREE_RUN = dict(base_pos=0.06, base_height=0.02, joint_pos=0.25)


def captures():
    return sorted(glob.glob(os.path.join(GOLD, "isaacgym_*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def dof_permutation(z):
    """column order that brings the capture's joints (Isaac Gym's actor dof order) into this build's (FL, FR, RL, RR) x (hip, thigh,
    calf): identity if the importer used the order SURVEY 8 assumed"""
    names = [str(n) for n in z["dof_names"]]
    mine = urdf_model.load_model("go1")["dof_names"]
    assert sorted(names) == sorted(mine), (names, mine)
    return [names.index(n) for n in mine]


def _to(engine, a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(engine.tensor(abi.T_ROOT_STATE).device)


def replay(engine_factory, path):
    """Returns {"one_step": {quantity: max error}, "free_run": {...}, "substeps": K}.  engine_factory(desc, keep) -> engine."""
    z, meta = load(path)
    task, N, A = meta["task"], int(meta["num_envs"]), int(meta["num_agents"])
    root, dof, tau = z["root"], z["dof"], z["tau"]
    K = tau.shape[0]
    perm = dof_permutation(z)
    cols = [r * 12 + p for r in range(A) for p in perm]
    dof = np.concatenate([dof[:, :, cols], dof[:, :, 12 * A:]], axis=2)
    tau = tau[:, :, cols]
    d, keep, ctx = make_desc(task, N)
    assert abs(float(z["sim_dt"]) - d.dt) < 1e-9
    assert np.allclose(ctx["env_origins"], z["env_origins"], atol=1e-4), "capture and build disagree on the env origins (terrain seed / track assignment)"
    e = engine_factory(d, keep)
    e.reset_all()
    R, D, T = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE), e.tensor(abi.T_TORQUES)
    assert tuple(R.shape) == root.shape[1:] and tuple(D.shape) == dof.shape[1:], (R.shape, root.shape, D.shape, dof.shape)
    sync = torch.cuda.synchronize if R.is_cuda else (lambda: None)

    def err(k, acc):
        sync()
        r, q = R.detach().cpu().numpy(), D.detach().cpu().numpy()
        acc["base_pos"] = max(acc.get("base_pos", 0.0), float(np.abs(r[:, :A, :3] - root[k][:, :A, :3]).max()))
        acc["base_height"] = max(acc.get("base_height", 0.0), float(np.abs(r[:, :A, 2] - root[k][:, :A, 2]).max()))
        acc["base_vel"] = max(acc.get("base_vel", 0.0), float(np.abs(r[:, :A, 7:10] - root[k][:, :A, 7:10]).max()))
        acc["joint_pos"] = max(acc.get("joint_pos", 0.0), float(np.abs(q[:, :12 * A, 0] - dof[k][:, :12 * A, 0]).max()))
        acc["joint_vel"] = max(acc.get("joint_vel", 0.0), float(np.abs(q[:, :12 * A, 1] - dof[k][:, :12 * A, 1]).max()))
    one, free = {}, {}
    for k in range(K):                                   # one-step replay
        R.copy_(_to(e, root[k])); D.copy_(_to(e, dof[k])); T.copy_(_to(e, tau[k]))
        e.simulate()
        err(k + 1, one)
    R.copy_(_to(e, root[0])); D.copy_(_to(e, dof[0]))
    for k in range(K):                                   # free-running replay
        T.copy_(_to(e, tau[k]))
        e.simulate()
        err(k + 1, free)
    e.close()
    return {"task": task, "substeps": K, "one_step": one, "free_run": free, "source": meta.get("source", "isaacgym")}


def check(res):
    bad = []
    for k, tol in TOL_ONE_STEP.items():
        if not res["one_step"][k] <= tol:
            bad.append(f"one-step {k}: {res['one_step'][k]:.4g} > {tol}")
    for k, tol in TOL_FREE_RUN.items():
        if not res["free_run"][k] <= tol:
            bad.append(f"free-running {k}: {res['free_run'][k]:.4g} > {tol}")
    return bad


def synthetic_capture(path, task="go1gate", N=3, steps=6):
    """A file in the capture's format produced by the CPU ORACLE instead of Isaac Gym (meta.source says so): what the replay tests
    run on when no real capture is present, so that the consumer of the format is exercised in every CI run.  It pins nothing."""
    from helpers import oracle_engine
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "tools"))
    from capture_isaacgym_trace import scripted_torques
    d, keep, ctx = make_desc(task, N)
    e = oracle_engine(d, keep)
    e.reset_all()
    R, D, T = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE), e.tensor(abi.T_TORQUES)
    A = d.num_agents
    R[:, :, 7:] = 0
    q0 = torch.tensor([d.default_dof_pos[j] for j in range(12)])
    D[:, :12 * A, 0] = q0.repeat(A); D[:, :, 1] = 0
    lim = torch.tensor([d.torque_limits[j] for j in range(12)])
    root, dof, tau = [R.numpy().copy()], [D.numpy().copy()], []
    for k in range(steps * d.decimation):
        t = scripted_torques(D[:, :12 * A, 0], D[:, :12 * A, 1], q0, k, d.dt, lim)
        T.copy_(t)
        e.simulate()
        tau.append(t.numpy().copy()); root.append(R.numpy().copy()); dof.append(D.numpy().copy())
    m = urdf_model.load_model("go1")
    meta = dict(task=task, seed=0, num_envs=N, num_agents=A, num_npcs=d.num_npcs, decimation=d.decimation, sim_dt=d.dt, source="synthetic: this build's CPU oracle, NOT Isaac Gym")
    rev = [r * 12 + (11 - i) for r in range(A) for i in range(12)]      # stored in ANOTHER joint order on purpose: the name-based permutation must undo it
    dof_c, tau_c = np.stack(dof), np.stack(tau)
    dof_c = np.concatenate([dof_c[:, :, rev], dof_c[:, :, 12 * A:]], axis=2)
    np.savez_compressed(path, root=np.stack(root), dof=dof_c, tau=tau_c[:, :, rev], cf=np.zeros((len(tau), N, 0, 3), np.float32),
                        dof_names=np.array(m["dof_names"][::-1]), body_names=np.array(m["reported_body_names"]),
                        env_origins=ctx["env_origins"], agent_origins=ctx["agent_origins"], sim_dt=np.float32(d.dt), meta=np.array(json.dumps(meta)))
    e.close()
    return path
