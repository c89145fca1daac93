"""Contact-rich scenarios for row H ("Next" 5): sustained multi-contact motion whose answers do not come from the build's own
derivation -- a trotting Go1 under a time-step sweep, a drop onto four feet, a box on a ramp at the friction limit (along the
friction frame's axis and across it), two robots colliding in free flight.  Engine-agnostic: every function takes `make_engine(d, keep)`
(the f64 / f32 oracle on the CPU, the HIP engine under -m gpu) and drives it substep by substep through the ABI the reference's
`Go1.step` loop uses (compute_torques -> simulate; go1.py:48-58).  The robots are held by the reference's own PD law (control type
"P", legged_robot.py:385): a continuous feedback law, so the time-step limit is well defined -- the actuator network's history inputs
are per-substep samples and would change the controller with dt."""
import os

import numpy as np
import torch

import rigid_ref as rr
from helpers import make_desc, perlin_terrain
from mqe.engine import abi

G = 9.81


def _np(t):
    return t.detach().cpu().numpy()


def _pd(d, kp, kd):
    d.control_type = abi.CTRL["P"]
    d.kp, d.kd, d.action_scale = kp, kd, 1.0


def _stand(e, d, N):
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    q0 = torch.tensor([d.default_dof_pos[j] for j in range(12)] * d.num_agents, device=root.device)
    dof[:, :12 * d.num_agents, 0] = q0
    dof[..., 1] = 0
    root[..., 7:] = 0
    e.tensor(abi.T_ACTIONS).zero_()
    return root, dof


def state_of(root, dof, env, r):
    return rr.split_state(root[env, r], dof[env, r * 12:(r + 1) * 12, 0], dof[env, r * 12:(r + 1) * 12, 1])


# ---- (a) a trotting robot under a time-step sweep ----------------------------------------------------------------------------------
def trot(make_engine, dt, T=0.5, N=8, seed=0, amp_scale=1.0):
    """N envs x 2 robots trot on the spot for T seconds under PD control towards a CONTINUOUS-time joint-target trajectory (diagonal leg
    pairs in phase; amplitude 0.15-0.4 rad, 1.5-3 Hz, drawn per robot from `seed`), from the default stance at rest.  Returns the final base
    positions / joint angles, the mean base height, the vertical contact impulse per robot and the momentum theorem's right-hand side
    m g T + P_z(T) - P_z(0) (momenta by tests/rigid_ref.py: float64 kinematics, differenced poses), and which robots fell."""
    d, k, _ = make_desc("go1gate", N)
    _pd(d, 20.0, 0.5)
    d.dt = dt
    e = make_engine(d, k)
    e.reset_all()
    root, dof = _stand(e, d, N)
    act = e.tensor(abi.T_ACTIONS)
    cf = e.tensor(abi.T_CONTACT_FORCE).reshape(N, 2, abi.NREP, 3)
    rng = np.random.RandomState(seed)
    amp = amp_scale * (0.15 + 0.25 * rng.rand(N, 2, 1))
    freq = 1.5 + 1.5 * rng.rand(N, 2, 1)
    ph0 = 2 * np.pi * rng.rand(N, 2, 1)
    legph = np.array([0, np.pi, np.pi, 0])[None, None, :]
    m = rr.load_model()
    r0, q0 = _np(root).astype(np.float64), _np(dof).astype(np.float64)
    Pz0 = np.array([[rr.momenta(m, *state_of(r0, q0, env, r))[2][2] for r in range(2)] for env in range(N)])
    n = int(round(T / dt))
    imp, hsum = np.zeros((N, 2)), np.zeros((N, 2))
    trunk = np.zeros((N, 2), bool)
    for s in range(n):
        ph = 2 * np.pi * freq * (s * dt) + ph0 + legph
        a = np.zeros((N, 2, 4, 3), np.float32)
        a[..., 0] = 0.1 * amp * np.cos(ph)
        a[..., 1] = amp * np.sin(ph)
        a[..., 2] = -1.6 * amp * np.sin(ph) + 0.3 * amp * np.cos(ph)
        act.copy_(torch.from_numpy(a.reshape(N, 24)))
        e.compute_torques()
        e.simulate()
        f = _np(cf)
        imp += f[..., 2].sum(-1) * dt
        trunk |= np.linalg.norm(f[:, :, 0], axis=-1) > 1.0
        hsum += _np(root)[:, :, 2]
    r1, q1 = _np(root).astype(np.float64), _np(dof).astype(np.float64)
    Pz1 = np.array([[rr.momenta(m, *state_of(r1, q1, env, r))[2][2] for r in range(2)] for env in range(N)])
    mt = float(m["mass"].sum())
    e.close() if hasattr(e, "close") else None
    return dict(pos=r1[:, :, :3], q=q1[:, :24, 0], height=hsum / n, impulse=imp, impulse_expected=mt * G * n * dt + (Pz1 - Pz0),
                fell=trunk | (r1[:, :, 2] < 0.18), mass=mt, steps=n)


# ---- (b1) dropped onto four feet ---------------------------------------------------------------------------------------------------
def drop(make_engine, height=0.05, T=1.0):
    """Two robots standing under PD control (settled first), lifted by `height` and released at rest.  Restitution 0: the fall's kinetic
    energy is gone after the landing and the feet carry m g again.  Returns kinetic-energy history (independent kinematics), the vertical
    foot-force sum at the end, the contact impulse against the momentum theorem, the feet's rebound and whether the trunk ever touched."""
    d, k, _ = make_desc("go1gate", 1)
    _pd(d, 40.0, 1.0)
    e = make_engine(d, k)
    e.reset_all()
    root, dof = _stand(e, d, 1)
    for s in range(200):
        e.compute_torques()
        e.simulate()
    zstand = _np(root)[0, :, 2].astype(np.float64)
    root[0, :, 2] += height
    root[..., 7:] = 0
    dof[..., 1] = 0
    cf = e.tensor(abi.T_CONTACT_FORCE).reshape(1, 2, abi.NREP, 3)
    m = rr.load_model()
    mt = float(m["mass"].sum())
    P0 = [rr.momenta(m, *state_of(_np(root).astype(np.float64), _np(dof).astype(np.float64), 0, r))[2] for r in range(2)]
    n = int(round(T / d.dt))
    imp, ke, fz, zz, trunk = np.zeros((2, 3)), [], [], [], 0.0
    for s in range(n):
        e.compute_torques()
        e.simulate()
        f = _np(cf)[0].astype(np.float64)
        imp += f.sum(1) * d.dt
        trunk = max(trunk, float(np.linalg.norm(f[:, 0], axis=-1).max()))
        rs, qs = _np(root).astype(np.float64), _np(dof).astype(np.float64)
        ke.append([rr.kinetic_energy(m, *state_of(rs, qs, 0, r)) for r in range(2)])
        fz.append(f[:, :, 2].sum(-1))
        zz.append(rs[0, :, 2].copy())
    P1 = [rr.momenta(m, *state_of(_np(root).astype(np.float64), _np(dof).astype(np.float64), 0, r))[2] for r in range(2)]
    ke, fz, zz = np.array(ke), np.array(fz), np.array(zz)
    first = int(np.argmax(fz[:, 0] > 0))
    bal = np.array([imp[r] - (P1[r] - P0[r]) - np.array([0, 0, mt * G * n * d.dt]) for r in range(2)]) / (mt * G * n * d.dt)
    e.close() if hasattr(e, "close") else None
    return dict(ke=ke, ke_fall=mt * G * height, fz_end=fz[-40:].mean(0), weight=mt * G, z_end=zz[-1] - zstand, impulse_balance=bal,
                rebound=float((zz[first:] - zz[first]).max()), first_contact_step=first, trunk_force=trunk)


# ---- (b2) a box on a ramp at the friction limit ------------------------------------------------------------------------------------
def friction_frame_limit(mu, diag):
    """tan(theta) up to which the friction BOX (two tangent rows, each bounded by mu lambda_n; tangent frame = contact_tangents of the
    normal: t1 = x cross n for |n_z| > 0.7) holds a body on a plane tilted along +y (diag = False: gravity's tangential part lies along
    t1 alone -> mu) or along the (1, 1) diagonal (it splits over t1 and t2 -> the larger share decides; fixed point of the slope).
    Computed here from the frame's definition, not from the engine."""
    if not diag:
        return mu
    s = mu
    for _ in range(50):
        n = np.array([-s / np.sqrt(2), -s / np.sqrt(2), 1.0])
        n /= np.linalg.norm(n)
        t1 = np.cross([1.0, 0, 0], n)
        t1 /= np.linalg.norm(t1)
        t2 = np.cross(n, t1)
        down = np.array([-1 / np.sqrt(2), -1 / np.sqrt(2), -s])
        down /= np.linalg.norm(down)
        s = mu / max(abs(down @ t1), abs(down @ t2))
    return float(s)


def box_on_ramp(make_engine, slope, diag, mu=0.5, T=0.5):
    """go1pushbox's free box (6 kg, box.urdf) resting flat on an inclined plane h = slope * (x dx + y dy) given as the relief map (exact under
    bilinear sampling), friction mu, released at rest; the robots are out of the way.  Returns the distance slid downhill, the downhill
    speed, the gap to the plane and the sideways drift after T seconds."""
    d, k, _ = make_desc("go1pushbox", 1, terrain_cfg=perlin_terrain("go1pushbox", zScale=0.01))
    hs = d.horizontal_scale
    X, Y = np.meshgrid(np.arange(d.sdf_nx) * hs, np.arange(d.sdf_ny) * hs, indexing="ij")
    dirv = np.array([1.0, 1.0]) / np.sqrt(2) if diag else np.array([0.0, 1.0])
    ramp = np.ascontiguousarray((slope * (X * dirv[0] + Y * dirv[1])).astype(np.float32))
    k.append(ramp)
    d.ground_height = ramp.ctypes.data_as(abi.FP)
    d.friction = mu
    e = make_engine(d, k)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    A = d.num_agents
    root[0, :A, 2] += 30.0
    hb = np.array([d.npc_box_half[i] for i in range(3)])
    n = np.array([-slope * dirv[0], -slope * dirv[1], 1.0])
    n /= np.linalg.norm(n)
    ax = np.cross([0, 0, 1.0], n)
    sn = np.linalg.norm(ax)
    ax = ax / sn if sn > 1e-12 else np.array([1.0, 0, 0])
    ang = np.arcsin(sn)
    quat = np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]])

    def above(c):
        return (c[2] - d.ground_z - slope * (c[0] * dirv[0] + c[1] * dirv[1])) * n[2]
    c = _np(root)[0, A, :3].astype(np.float64)
    c[2] = d.ground_z + slope * (c[0] * dirv[0] + c[1] * dirv[1])
    c = c + n * (hb[2] - above(c))
    root[0, A, :3] = torch.tensor(c, dtype=torch.float32)
    root[0, A, 3:7] = torch.tensor(quat, dtype=torch.float32)
    root[0, A, 7:] = 0
    c0 = _np(root)[0, A, :3].astype(np.float64)
    for t in range(int(round(T / d.dt))):
        e.simulate()
    c1 = _np(root)[0, A, :3].astype(np.float64)
    down = np.array([-dirv[0], -dirv[1], -slope])
    down /= np.linalg.norm(down)
    slid = float((c1 - c0) @ down)
    out = dict(slid=slid, speed=float(_np(root)[0, A, 7:10].astype(np.float64) @ down), gap=float(above(c1) - hb[2]),
               sideways=float(np.linalg.norm((c1 - c0) - slid * down - ((c1 - c0) @ n) * n)), theta=float(np.arctan(slope)))
    e.close() if hasattr(e, "close") else None
    return out


# ---- (b3) two robots collide in free flight ------------------------------------------------------------------------------------------
def collide_in_flight(make_engine, T=0.5, gap=0.55, speed=1.0, dt=None):
    """Zero gravity, two robots held in their stance by PD control drift into each other (tumbling slowly): every force is internal
    to the pair, so the pair's linear momentum and its angular momentum about the origin do not change -- while each robot's own
    momentum does.  Momenta from tests/rigid_ref.py."""
    d, k, _ = make_desc("go1gate", 1)
    _pd(d, 40.0, 1.0)
    d.gravity_z = 0.0
    if dt is not None:
        d.dt = dt
    e = make_engine(d, k)
    e.reset_all()
    root, dof = _stand(e, d, 1)
    root[0, :, 2] = 20.0
    y = float(root[0, 0, 1])
    root[0, 0, 1] = y
    root[0, 1, 1] = y + gap
    root[0, 1, 0] = root[0, 0, 0] + 0.07
    root[0, 0, 8] = +0.5 * speed
    root[0, 1, 8] = -0.5 * speed
    root[0, 0, 10:13] = torch.tensor([0.3, -0.2, 0.4])
    root[0, 1, 10:13] = torch.tensor([-0.2, 0.3, 0.1])
    m = rr.load_model()

    X0 = _np(root)[0, :, :3].astype(np.float64).mean(0)          # a FIXED point near the pair (about the far-away origin C x P would hide everything)

    def totals():
        rs, qs = _np(root).astype(np.float64), _np(dof).astype(np.float64)
        P, L, each = np.zeros(3), np.zeros(3), []
        for r in range(2):
            mt, C, Pr, Lr = rr.momenta(m, *state_of(rs, qs, 0, r))
            P += Pr
            L += Lr + np.cross(C - X0, Pr)
            each.append(Pr)
        return P, L, np.array(each)
    P0, L0, e0 = totals()
    cf = e.tensor(abi.T_CONTACT_FORCE)
    touched = 0.0
    n = int(round(T / d.dt))
    for s in range(n):
        e.compute_torques()
        e.simulate()
        touched = max(touched, float(np.abs(_np(cf)).max()))
    P1, L1, e1 = totals()
    e.close() if hasattr(e, "close") else None
    return dict(dP=P1 - P0, dL=L1 - L0, dP_each=e1 - e0, max_contact_force=touched, mass=float(m["mass"].sum()), L0=L0)


