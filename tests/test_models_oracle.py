"""MLPs of the path (rows C, D, F) and the URDF-derived model, CPU oracle vs the reference's golden vectors."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import golden, make_desc, oracle_engine
from mqe.utils import urdf_model


def test_actuator_net_matches_torchscript():
    z = golden("actuator_net")
    d, k, _ = make_desc("go1gate", 1)
    e = oracle_engine(d, k)
    f = e.lib.mqo_actuator_net
    f.argtypes, f.restype = [C.c_void_p, C.c_void_p], C.c_float
    for x, y in ((z["x"], z["y"]), (z["spot_x"], z["spot_y"])):
        rows = [np.ascontiguousarray(r, np.float32) for r in x]          # keep alive across the foreign calls
        got = np.array([f(e.h, r.ctypes.data) for r in rows])
        np.testing.assert_allclose(got, y, rtol=2e-5, atol=2e-5)
    assert abs(z["spot_y"][0] - 19.5816) < 1e-3          # SURVEY 8(c) spot value


def test_adaptation_module_matches_torchscript():
    z = golden("adaptation_module")
    d, k, _ = make_desc("go1gate", 1)
    e = oracle_engine(d, k)
    f = e.lib.mqo_policy_forward
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lat, act = np.zeros(2, np.float32), np.zeros(12, np.float32)
    for i in range(z["x"].shape[0]):
        xi = np.ascontiguousarray(z["x"][i], np.float32)
        f(e.h, xi.ctypes.data, lat.ctypes.data, act.ctypes.data)
        np.testing.assert_allclose(lat, z["y"][i], rtol=1e-4, atol=2e-5)
    zero = np.zeros(2100, np.float32)
    f(e.h, zero.ctypes.data, lat.ctypes.data, act.ctypes.data)
    np.testing.assert_allclose(lat, [7.8227, 0.8373], atol=1e-3)   # f(0) known answer


def test_go1_model_from_urdf_numbers():
    m = urdf_model.load_model("go1")
    assert m["dof_names"][:3] == ["FL_hip_joint", "FL_thigh_joint", "FL_calf_joint"] and m["dof_names"][9] == "RR_hip_joint"
    assert len(m["reported_body_names"]) == 17 and m["reported_body_names"][4] == "FL_foot"
    # trunk 4.8 + imu 0.001; hip .510299; thigh .898919; calf .158015 + foot .06  (go1.urdf inertial blocks)
    np.testing.assert_allclose(m["mass"][0], 4.801, atol=1e-9)
    np.testing.assert_allclose(m["mass"][3], 0.158015 + 0.06, atol=1e-9)
    np.testing.assert_allclose(m["total_mass"], 4.801 + 4 * (0.510299 + 0.898919 + 0.218015), atol=1e-9)
    assert m["parent"] == [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11]
    # collision model: the URDF's 18 primitives (2 boxes on the base; hip capsule, thigh and calf capsules, foot sphere per leg) and
    # their 32 feature points (4 feet, 8 trunk + 4 head corners, 4 knees, 4 thigh tops, 8 hip capsule ends)
    assert len(m["prim_type"]) == 18 and sorted(m["prim_type"]) == [0] * 4 + [1] * 12 + [2] * 2
    assert len(m["sphere_body"]) == 32 and m["sphere_radius"][0] == 0.02 and m["sphere_tag"][:4] == ["foot"] * 4
    assert all(m["prim_body"][m["sphere_prim"][i]] == m["sphere_body"][i] for i in range(32))
    for I in m["inertia"]:
        assert np.all(np.linalg.eigvalsh(np.asarray(I)) > 0)


def scripted_body(tmp_path, hidden=(128, 64), seed=5):
    """A TorchScript body as upstream ships it (Sequential of Linear / ELU, go1.py:397) with hidden sizes that are NOT the
    stand-in's 512-256-128: written to <tmp>/body_latest.jit, read back through the product loader (policy_weights.load_body)."""
    from mqe.utils import policy_weights as pw
    torch.manual_seed(seed)
    dims = (pw.BODY_IN,) + tuple(hidden) + (pw.BODY_OUT,)
    layers = []
    for i in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.mul_(2.0 if i else 1.0)
            lin.bias.uniform_(-0.3, 0.3)
        layers.append(lin)
        if i < len(dims) - 2:
            layers.append(torch.nn.ELU())
    net = torch.nn.Sequential(*layers).eval()
    torch.jit.script(net).save(str(tmp_path / "body_latest.jit"))
    Ws, bs, synthetic = pw.load_body(str(tmp_path))
    assert synthetic is False and [w.shape for w in Ws] == [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
    for w, b, lin in zip(Ws, bs, [l for l in layers if isinstance(l, torch.nn.Linear)]):
        assert np.array_equal(w, lin.weight.detach().numpy()) and np.array_equal(b, lin.bias.detach().numpy())
    return net, Ws, bs


def torch_adaptation():
    from mqe.utils import policy_weights as pw
    Ws, bs = pw.load_adaptation_module()
    layers = []
    for i, (w, b) in enumerate(zip(Ws, bs)):
        lin = torch.nn.Linear(w.shape[1], w.shape[0])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w)); lin.bias.copy_(torch.from_numpy(b))
        layers.append(lin)
        if i < len(Ws) - 1:
            layers.append(torch.nn.ELU())
    return torch.nn.Sequential(*layers).eval()


def test_real_body_file_of_another_shape_runs_through_the_loader(tmp_path):
    """Row D: body_latest.jit is missing upstream, so what can be pinned is the path a real file takes: TorchScript ->
    load_body -> descriptor -> oracle forward == the TorchScript module's own output on the same inputs (plain torch fp32),
    for a network whose depth and widths differ from the synthetic stand-in (2102-128-64-12)."""
    net, Ws, bs = scripted_body(tmp_path)
    ada = torch_adaptation()
    d, k, _ = make_desc("go1gate", 1, body=(Ws, bs))
    assert [d.body.dims[i] for i in range(d.body.n_layers + 1)] == [2102, 128, 64, 12]
    e = oracle_engine(d, k)
    f = e.lib.mqo_policy_forward
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    g = torch.Generator().manual_seed(2)
    lat, act = np.zeros(2, np.float32), np.zeros(12, np.float32)
    for i in range(8):
        x = torch.randn(2100, generator=g) * (0.3 if i else 0.0)
        with torch.no_grad():
            lt = ada(x[None])
            at = net(torch.cat([x[None], lt], dim=1))
        xi = np.ascontiguousarray(x.numpy(), np.float32)
        f(e.h, xi.ctypes.data, lat.ctypes.data, act.ctypes.data)
        np.testing.assert_allclose(lat, lt[0].numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(act, at[0].numpy(), rtol=1e-4, atol=5e-5)


# ---- the URDF reader that feeds BOTH engines, against an independent reading of go1.urdf (tests/golden/go1_urdf_facts.json) -----------
def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _urdf_facts():
    import json
    import os
    from helpers import GOLD
    return json.load(open(os.path.join(GOLD, "go1_urdf_facts.json")))


def _welded(f, body):
    """links rigidly attached to `body` through fixed joints: [(link name, R, t) in the body's frame]"""
    out, todo = [], [(body, np.eye(3), np.zeros(3))]
    while todo:
        name, R, t = todo.pop()
        out.append((name, R, t))
        for j in f["joints"].values():
            if j["parent"] == name and j["type"] == "fixed":
                todo.append((j["child"], R @ _rpy(*j["rpy"]), R @ np.array(j["xyz"]) + t))
    return out


def test_model_file_agrees_with_an_independent_reading_of_the_urdf():
    """assets/go1_model.json (mqe/utils/urdf_model.py: fixed-joint collapsing, inertia merging, joint table) feeds the oracle AND the
    HIP engine, so a reading error would be common-mode.  Here every dynamic body is rebuilt from the raw URDF numbers with this
    test's own parallel-axis arithmetic: mass, centre of mass, inertia about it, joint origin / axis / limits / speed / effort."""
    f = _urdf_facts()
    m = urdf_model.load_model("go1")
    assert m["body_names"][0] == "base" and m["body_names"][1:4] == ["FL_hip", "FL_thigh", "FL_calf"]     # base -(fixed)-> trunk, imu, head
    for b, name in enumerate(m["body_names"]):
        M, mc = 0.0, np.zeros(3)
        parts = []
        for ln, R, t in _welded(f, name):
            L = f["links"][ln]
            if L.get("mass", 0) > 0:
                c = R @ np.array(L["com_xyz"]) + t
                ixx, ixy, ixz, iyy, iyz, izz = L["inertia"]
                Rl = R @ _rpy(*L["com_rpy"])
                I = Rl @ np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]) @ Rl.T
                parts.append((L["mass"], c, I))
                M += L["mass"]; mc += L["mass"] * c
        com = mc / M
        I = sum(Ii + mi * ((ci - com) @ (ci - com) * np.eye(3) - np.outer(ci - com, ci - com)) for mi, ci, Ii in parts)
        assert abs(m["mass"][b] - M) < 1e-12
        np.testing.assert_allclose(m["com"][b], com, atol=1e-12)
        np.testing.assert_allclose(np.asarray(m["inertia"][b]), I, atol=1e-12)
        if b:
            j = f["joints"][m["dof_names"][b - 1]]
            assert j["type"] == "revolute" and j["child"] == name
            weld = {ln: (R, t) for ln, R, t in _welded(f, m["body_names"][m["parent"][b]])}      # the joint's parent link is welded into the parent body
            Rp, tp = weld[j["parent"]]
            np.testing.assert_allclose(m["joint_offset"][b], Rp @ np.array(j["xyz"]) + tp, atol=1e-15)
            np.testing.assert_allclose(m["joint_axis"][b], j["axis"], atol=0)
            assert np.allclose(Rp, np.eye(3))
            lim = j["limit"]
            assert (m["dof_lower"][b - 1], m["dof_upper"][b - 1], m["dof_velocity"][b - 1], m["dof_effort"][b - 1]) == (lim["lower"], lim["upper"], lim["velocity"], lim["effort"])
            assert j["rpy"] == [0.0, 0.0, 0.0]
    # every revolute joint of the file is a dof of the model, every massive link ends up in exactly one body
    assert sorted(m["dof_names"]) == sorted(k for k, j in f["joints"].items() if j["type"] == "revolute")
    in_bodies = [ln for name in m["body_names"] for ln, _, _ in _welded(f, name)]
    assert len(in_bodies) == len(set(in_bodies))
    assert abs(m["total_mass"] - sum(f["links"][ln].get("mass", 0.0) for ln in in_bodies)) < 1e-12


def _surface_points(c, rng, n=400):
    """points on the surface of a URDF collision primitive, in the link frame of the primitive's owner"""
    R, t = _rpy(*c["rpy"]), np.array(c["xyz"])
    if c["type"] == "sphere":
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        p = v * c["params"]["radius"][0]
    elif c["type"] == "box":
        h = np.array(c["params"]["size"]) / 2
        p = rng.uniform(-1, 1, (n, 3)) * h
        ax = rng.integers(0, 3, n)
        p[np.arange(n), ax] = np.sign(rng.uniform(-1, 1, n)) * h[ax]
    else:   # cylinder along z
        r, L = c["params"]["radius"][0], c["params"]["length"][0]
        a = rng.uniform(0, 2 * np.pi, n)
        p = np.stack([r * np.cos(a), r * np.sin(a), rng.uniform(-L / 2, L / 2, n)], 1)
        cap = rng.uniform(0, 1, n) < 0.3
        rr = r * np.sqrt(rng.uniform(0, 1, n))
        p[cap] = np.stack([rr * np.cos(a), rr * np.sin(a), np.sign(rng.uniform(-1, 1, n)) * L / 2], 1)[cap]
    return p @ R.T + t


def _prim_sdf(c, x):
    """signed distance of points x (link frame) to the primitive"""
    R, t = _rpy(*c["rpy"]), np.array(c["xyz"])
    q = (x - t) @ R
    if c["type"] == "sphere":
        return np.linalg.norm(q, axis=1) - c["params"]["radius"][0]
    if c["type"] == "box":
        d = np.abs(q) - np.array(c["params"]["size"]) / 2
        return np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(d.max(1), 0)
    r, L = c["params"]["radius"][0], c["params"]["length"][0]
    d = np.stack([np.linalg.norm(q[:, :2], axis=1) - r, np.abs(q[:, 2]) - L / 2], 1)
    return np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(d.max(1), 0)


def _model_prim_sdf(m, q, Rb, pb, x):
    """signed distance of world points x to primitive q of the engine's collision model (sphere / capsule / link-aligned box)"""
    b = m["prim_body"][q]
    c = pb[b] + Rb[b] @ np.array(m["prim_center"][q])
    if m["prim_type"][q] == 2:
        d = np.abs((x - c) @ Rb[b]) - np.array(m["prim_half"][q])
        return np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(d.max(1), 0)
    u = Rb[b] @ np.array(m["prim_axis"][q])
    uu = max(float(u @ u), 1e-30)
    t = np.clip(((x - c) @ u) / uu, -1, 1)
    return np.linalg.norm(x - c - t[:, None] * u, axis=1) - m["prim_half"][q][0]


def _model_prim_surface(m, q, Rb, pb, rng, n=400):
    b = m["prim_body"][q]
    c = pb[b] + Rb[b] @ np.array(m["prim_center"][q])
    if m["prim_type"][q] == 2:
        h = np.array(m["prim_half"][q])
        p = rng.uniform(-1, 1, (n, 3)) * h
        ax = rng.integers(0, 3, n)
        p[np.arange(n), ax] = np.sign(rng.uniform(-1, 1, n)) * h[ax]
        return c + p @ Rb[b].T
    u = Rb[b] @ np.array(m["prim_axis"][q])
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    t = rng.uniform(-1, 1, n)
    if float(u @ u) > 0:                                    # capsule: the end caps' hemispheres, else the cylinder wall
        ul = u / np.linalg.norm(u)
        cap = rng.uniform(0, 1, n) < 0.4
        wall = v - np.outer(v @ ul, ul); wall /= np.linalg.norm(wall, axis=1, keepdims=True)
        sgn = np.sign(v @ ul); sgn[sgn == 0] = 1
        return np.where(cap[:, None], c + sgn[:, None] * u + m["prim_half"][q][0] * v, c + t[:, None] * u + m["prim_half"][q][0] * wall)
    return c + m["prim_half"][q][0] * v


@pytest.mark.parametrize("variant", ["capsule", "exact"])
def test_collision_model_stays_close_to_the_urdf_collision_primitives(variant):
    """go1.urdf declares 2 boxes on the base (:56, :80), a cylinder per hip (-> capsule, go1_config.py:75), a box per thigh and calf
    and a sphere per foot.  The engine collides those primitives themselves, except that the thigh / calf bars are capsules
    (VERDICT r2 missing #2).  Two-sided bound on the WHOLE robot, default stance and random poses: (a) how far the URDF primitives'
    outer surface sticks out of the model's primitives and vice versa -- only the bars' edges and end corners differ; (b) the same
    for the support function h(d) = max x . d over the model's FEATURE POINTS (all a plane -- ground, wall face, box face, plank --
    ever sees): a convex body's support point is a feature point, so it is the primitives' own up to the bars' cross-section."""
    import rigid_ref as rr
    f = _urdf_facts()
    m = urdf_model.load_model("go1")
    if variant == "exact":        # desc collision_model = "exact": thigh / calf as the URDF's boxes, 60 feature points (round 4)
        m = dict(m, **m["exact"])
    mr = rr.load_model()
    rng = np.random.default_rng(0)
    q_def = np.array([0.1, 0.8, -1.5, -0.1, 0.8, -1.5, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5])
    lo = np.array([-0.8, -1.0, -2.6] * 4); hi = np.array([0.8, 4.0, -0.95] * 4)
    res = []
    for q in [q_def] + [lo + (hi - lo) * rng.uniform(0.25, 0.75, 12) for _ in range(3)]:
        Rb, pb = rr.fk(mr, np.zeros(3), np.eye(3), q)
        feat = [(pb[m["sphere_body"][i]] + Rb[m["sphere_body"][i]] @ np.array(m["sphere_center"][i]), m["sphere_radius"][i]) for i in range(len(m["sphere_body"]))]
        prims = []
        for b, name in enumerate(m["body_names"]):
            for ln, R, t in _welded(f, name):
                for c in f["links"][ln]["collisions"]:
                    prims.append(dict(c, _R=Rb[b] @ R @ _rpy(*c["rpy"]), _t=pb[b] + Rb[b] @ (R @ np.array(c["xyz"]) + t)))

        def sdf_urdf(x):
            out = np.full(len(x), 1e9)
            for c in prims:
                # a URDF cylinder is a capsule in the simulator (replace_cylinder_with_capsule): segment = its axis, radius = its radius
                if c["type"] == "cylinder":
                    r, L = c["params"]["radius"][0], c["params"]["length"][0]
                    ql = (x - c["_t"]) @ c["_R"]
                    ql[:, 2] -= np.clip(ql[:, 2], -L / 2, L / 2)
                    out = np.minimum(out, np.linalg.norm(ql, axis=1) - r)
                else:
                    out = np.minimum(out, _prim_sdf(dict(c, rpy=[0, 0, 0], xyz=[0, 0, 0]), (x - c["_t"]) @ c["_R"]))
            return out

        def sdf_model(x):
            return np.min([_model_prim_sdf(m, qq, Rb, pb, x) for qq in range(len(m["prim_type"]))], axis=0)
        pts_u = []
        for c in prims:
            if c["type"] == "cylinder":                         # sample the capsule the simulator makes of it
                r, L = c["params"]["radius"][0], c["params"]["length"][0]
                v = rng.normal(size=(400, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
                z = rng.uniform(-L / 2, L / 2, 400)
                cap = rng.uniform(0, 1, 400) < 0.5
                wall = v.copy(); wall[:, 2] = 0; wall /= np.linalg.norm(wall, axis=1, keepdims=True)
                pl = np.where(cap[:, None], r * v + np.stack([0 * z, 0 * z, np.sign(v[:, 2]) * L / 2], 1), r * wall + np.stack([0 * z, 0 * z, z], 1))
                pts_u.append(pl @ c["_R"].T + c["_t"])
            else:
                pts_u.append(_surface_points(dict(c, rpy=[0, 0, 0], xyz=[0, 0, 0]), rng) @ c["_R"].T + c["_t"])
        pts_u = np.concatenate(pts_u)
        pts_u = pts_u[sdf_urdf(pts_u) > -1e-9]                  # the union's outer surface only
        pts_m = np.concatenate([_model_prim_surface(m, qq, Rb, pb, rng) for qq in range(len(m["prim_type"]))])
        pts_m = pts_m[sdf_model(pts_m) > -1e-9]
        dirs = rng.normal(size=(3000, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        hp = (pts_u @ dirs.T).max(0)
        if variant == "exact":        # the URDF union's support function in closed form (the sampled one is short by up to a few mm)
            sup = []
            for c in prims:
                dl = dirs @ c["_R"]
                if c["type"] == "box":
                    sup.append(dirs @ c["_t"] + np.abs(dl) @ (np.array(c["params"]["size"]) / 2))
                elif c["type"] == "sphere":
                    sup.append(dirs @ c["_t"] + c["params"]["radius"][0])
                else:
                    sup.append(dirs @ c["_t"] + np.abs(dl[:, 2]) * c["params"]["length"][0] / 2 + c["params"]["radius"][0])
            hp = np.max(sup, axis=0)
        hs = np.max([c @ dirs.T + r for c, r in feat], axis=0)
        down = dirs[:, 2] < -0.8                                # towards the ground
        res.append(dict(surface_out=float(sdf_model(pts_u).max()), surface_in=float(sdf_urdf(pts_m).max()),
                        support_out=float((hp - hs).max()), support_in=float((hs - hp).max()),
                        ground_out=float((hp - hs)[down].max()), ground_in=float((hs - hp)[down].max())))
    worst = {k: max(r[k] for r in res) for k in res[0]}
    print("collision-model bounds [m]:", variant, {k: round(v, 5) for k, v in worst.items()})
    if variant == "exact":
        # every primitive is the URDF's own shape (surface deviation: rounding) and the feature points carry the
        # union's support function to 1 mm in every direction (what is left out: corners that lie inside a neighbouring shape)
        assert max(worst.values()) < 1e-3, worst
        return
    # towards the ground the robot is its feet: the URDF's own spheres -> exact (sampling noise of the surface points only)
    assert worst["ground_out"] < 2e-3 and worst["ground_in"] < 2e-3, worst
    # any plane: the feature points reach as far as the primitives to 6 mm (a thigh bar's end corner against the round cap of its
    # capsule) and nowhere more than 6.6 mm farther (the cap's tip beyond the bar's end face): the optimum of a capsule against a
    # 213 x 24.5 x 34 mm bar (mqe/utils/urdf_model.py::_capsule_for_bar); measured 5.6 / 6.0 mm (round 2's sphere sets: 13 / 7 mm)
    assert worst["support_out"] < 0.0065 and worst["support_in"] < 0.007, worst
    # the whole surface (what an edge or another body can touch): <= 6.5 mm either way, all of it on the thigh bars' edges and end
    # corners; trunk and head boxes, hip capsules and feet are the URDF's own shapes (round 2: 44 / 21 mm)
    assert worst["surface_out"] < 0.007 and worst["surface_in"] < 0.007, worst
    # the bodies that carry the robot in every task -- the feet -- are the URDF's own spheres
    for leg in ("FL", "FR", "RL", "RR"):
        foot = f["links"][leg + "_foot"]["collisions"][0]
        assert foot["type"] == "sphere" and foot["params"]["radius"][0] in [m["sphere_radius"][i] for i in range(4)]


def test_fma_clone_of_the_policy_layers_is_bit_identical(tmp_path, monkeypatch):
    """oracle/mqe_oracle.c builds matvec_chain twice (target_clones: FMA3 / default) and the loader picks one by CPUID: a fused multiply-add is
    correctly rounded whether libm or the vfmadd instruction computes it, so a build WITHOUT the clone must give the same bits."""
    import os
    import subprocess
    import torch
    from helpers import make_desc, oracle_engine
    from mqe.engine import abi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "libmqe_oracle_noclone.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-std=gnu11", "-DMQO_NO_FMA_CLONE", "-w", "-o", lib,
                           os.path.join(root, "oracle", "mqe_oracle.c"), "-lm"])
    outs = []
    for which in (None, lib):
        if which:
            monkeypatch.setenv("MQE_ORACLE_LIB", which)
        d, k, _ = make_desc("go1gate", 6)
        e = oracle_engine(d, k)
        e.reset_all()
        g = torch.Generator().manual_seed(5)
        for t in range(35):                                  # past the 30-frame history horizon: every column of layer 0 carries data
            e.step(torch.rand(6, 2, 3, generator=g) * 2 - 1)
        outs.append([e.tensor(t).clone() for t in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_TORQUES, abi.T_WRAPPER_OBS, abi.T_ACTIONS)])
        e.close()
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
