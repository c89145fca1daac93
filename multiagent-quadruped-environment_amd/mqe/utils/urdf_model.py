"""URDF -> flat rigid-body model arrays consumed by the HIP engine (and by the test oracle).

Scope: the assets on the step() hot path -- `resources/robots/go1/urdf/go1.urdf` and the NPC objects
`resources/objects/{ball,sheep,seesaw}.urdf` of the reference (SURVEY 8c "files a CPU restatement of the
physics must follow").  What Isaac Gym's importer does with them is restated here:

* `collapse_fixed_joints=True` (reference go1_config.py:70): links joined by fixed joints are merged into their
  parent (mass, COM, inertia, collision shapes), except joints tagged dont_collapse="true"
  (go1.urdf:207,330,453,576 - the feet), which stay *reported* bodies (17 per Go1) although they are
  dynamically welded to the calf.
* DOF / body order: per leg (hip, thigh, calf[, foot]) with legs ordered FL, FR, RL, RR.  The reference relies on
  hips at DOF 0,3,6,9 (go1.py:331) and looks default angles up by name (legged_robot.py:628-633); the leg order
  itself comes from Isaac Gym and is not stated in the reference, so the build fixes this one.
* `replace_cylinder_with_capsule=True` (go1_config.py:75).

Collision geometry is reduced to spheres rigidly attached to links (sphere-swept approximations of the URDF
primitives, rule in `_spheres_for_go1`); see DESIGN.md "collision model" for why (one closed-form narrow phase).
"""
import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

LEG_ORDER = ("FL", "FR", "RL", "RR")
MAX_SPHERES = 32
N_BODIES_DYN = 13       # base + 4 x (hip, thigh, calf)
N_BODIES_REPORTED = 17  # + 4 feet
N_DOF = 12


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n
    return np.array(v, np.float64)


def _rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(el):
    o = el.find("origin") if el is not None else None
    xyz = _vec(o.get("xyz", "0 0 0")) if o is not None else np.zeros(3)
    rpy = _vec(o.get("rpy", "0 0 0")) if o is not None else np.zeros(3)
    return _rpy_to_mat(rpy), xyz


def _stl_aabb(path):
    """(half extents, centre, n_triangles) of a binary STL's bounding box.  The bridge / wrestling scenery ships as
    SolidWorks STL exports that are plain boxes (12 triangles) plus thin painted rings; the box is what collides."""
    import struct
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    v = np.frombuffer(b, dtype=np.uint8, count=n * 50, offset=84).reshape(n, 50)[:, 12:48].copy().view(np.float32).reshape(n * 3, 3)
    lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
    return (hi - lo) / 2, (hi + lo) / 2, n


class Link:
    def __init__(self, el, base_dir=None):
        self.name = el.get("name")
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))
        ine = el.find("inertial")
        if ine is not None:
            R, t = _origin(ine)
            self.mass = float(ine.find("mass").get("value"))
            i = ine.find("inertia")
            g = lambda k: float(i.get(k, 0.0))  # noqa: E731
            I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
            self.com = t
            self.inertia = R @ I @ R.T
        self.shapes = []  # (kind, params, R, t)
        for c in el.findall("collision"):
            R, t = _origin(c)
            geo = c.find("geometry")
            ch = list(geo)[0]
            if ch.tag == "box":
                self.shapes.append(("box", _vec(ch.get("size")) / 2, R, t))
            elif ch.tag == "sphere":
                self.shapes.append(("sphere", float(ch.get("radius")), R, t))
            elif ch.tag == "cylinder":
                self.shapes.append(("cylinder", (float(ch.get("radius")), float(ch.get("length"))), R, t))
            elif ch.tag == "mesh" and base_dir is not None:
                half, ctr, ntri = _stl_aabb(os.path.normpath(os.path.join(base_dir, ch.get("filename"))))
                self.shapes.append(("box", half, R, R @ ctr + t))          # bounding box of the mesh, in the link frame


class Joint:
    def __init__(self, el):
        self.name = el.get("name")
        self.type = el.get("type")
        self.parent = el.find("parent").get("link")
        self.child = el.find("child").get("link")
        self.R, self.t = _origin(el)
        ax = el.find("axis")
        self.axis = _vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0, 0])
        self.dont_collapse = el.get("dont_collapse", "false") == "true"
        lim = el.find("limit")
        self.lower = float(lim.get("lower", 0.0)) if lim is not None else 0.0
        self.upper = float(lim.get("upper", 0.0)) if lim is not None else 0.0
        self.effort = float(lim.get("effort", 0.0)) if lim is not None else 0.0
        self.velocity = float(lim.get("velocity", 0.0)) if lim is not None else 0.0


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {l.get("name"): Link(l, os.path.dirname(path)) for l in root.findall("link")}
    joints = [Joint(j) for j in root.findall("joint")]
    children = {j.child for j in joints}
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, roots
    return links, joints, roots[0]


def _merge_inertial(m1, c1, I1, m2, c2, I2):
    """Combine two rigid inertials expressed in the same frame (parallel-axis theorem)."""
    m = m1 + m2
    if m == 0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    c = (m1 * c1 + m2 * c2) / m

    def shift(I, mm, d):
        return I + mm * (d @ d * np.eye(3) - np.outer(d, d))
    return m, c, shift(I1, m1, c1 - c) + shift(I2, m2, c2 - c)


class Body:
    """A dynamic body after fixed-joint collapsing, expressed in its own link frame."""

    def __init__(self, name):
        self.name = name
        self.mass, self.com, self.inertia = 0.0, np.zeros(3), np.zeros((3, 3))
        self.shapes = []     # (kind, params, R, t, reported_name)
        self.parent = -1
        self.joint = None    # Joint to parent (revolute) or None for the root
        self.reported = [name]  # reported rigid bodies welded into this one (itself first)


def collapse(links, joints, root):
    """Depth-first collapse; returns list of Body in (root, then children in file order)."""
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    bodies = []

    def absorb(body, link_name, R, t, reported_name):
        l = links[link_name]
        if l.mass > 0:
            body.mass, body.com, body.inertia = _merge_inertial(body.mass, body.com, body.inertia,
                                                                l.mass, R @ l.com + t, R @ l.inertia @ R.T)
        for kind, prm, Rs, ts in l.shapes:
            body.shapes.append((kind, prm, R @ Rs, R @ ts + t, reported_name))
        for j in by_parent.get(link_name, []):
            Rj, tj = R @ j.R, R @ j.t + t
            if j.type == "fixed":
                rep = reported_name
                if j.dont_collapse:
                    rep = j.child
                    body.reported.append(j.child)
                absorb(body, j.child, Rj, tj, rep)
            else:
                nb = Body(j.child)
                nb.parent = bodies.index(body)
                nb.joint = j
                nb.joint_R, nb.joint_t = Rj, tj
                bodies.append(nb)
                absorb(nb, j.child, np.eye(3), np.zeros(3), j.child)

    b0 = Body(root)
    bodies.append(b0)
    absorb(b0, root, np.eye(3), np.zeros(3), root)
    return bodies


def _spheres_for_go1(bodies, reported_names):
    """Sphere-swept stand-ins for the Go1 collision primitives (go1.urdf:56,80 boxes on base, hip cylinder,
    thigh/calf boxes, foot sphere).  Returns list of (dyn_body, centre, radius, reported_body_index)."""
    out = []
    for bi, b in enumerate(bodies):
        for kind, prm, R, t, rep in b.shapes:
            ri = reported_names.index(rep)
            if kind == "sphere":
                out.append((bi, t, prm, ri))
            elif kind == "cylinder":
                r, L = prm
                out.append((bi, t, r, ri))           # half-length 0.02 < r/2: one sphere
            elif kind == "box":
                h = np.asarray(prm)
                order = np.argsort(h)
                r = float(h[order[0]])
                long_ax = int(order[2])
                ext = float(h[long_ax]) - r
                if h[order[1]] > 2.0 * r:          # plate-like (head box 0.02 x 0.05 x 0.05): 2 x 2 grid
                    a1, a2 = int(order[1]), int(order[2])
                    for s1 in (-1, 1):
                        for s2 in (-1, 1):
                            c = np.zeros(3)
                            c[a1] = s1 * (h[a1] - r)
                            c[a2] = s2 * (h[a2] - r)
                            out.append((bi, R @ c + t, r, ri))
                elif b.parent < 0:                   # trunk box: three spheres along the long axis
                    r2 = float(h[order[1]]) if h[order[1]] < 1.3 * r else r
                    for s in (-1, 0, 1):
                        c = np.zeros(3)
                        c[long_ax] = s * (h[long_ax] - r2)
                        out.append((bi, R @ c + t, r2, ri))
                else:                                # thigh / calf bars
                    rr = float(h[order[1]])          # larger of the two short half-extents
                    is_calf = "calf" in b.name
                    pts = (0.0,) if is_calf else (0.0, -1.0)   # calf: middle (ends = knee sphere of thigh, foot)
                    for s in pts:
                        c = np.zeros(3)
                        c[long_ax] = s * (h[long_ax] - rr)
                        # long axis of the bar maps to -z of the link (rpy 0,pi/2,0): "-1" end must be the far end
                        p = R @ c + t
                        if s != 0.0 and np.linalg.norm(p) < np.linalg.norm(t):
                            p = R @ (-c) + t
                        out.append((bi, p, rr, ri))
    return out


def build_go1_model(urdf_path):
    links, joints, root = parse_urdf(urdf_path)
    bodies = collapse(links, joints, root)
    # reorder legs FL, FR, RL, RR
    base = bodies[0]
    legs = {}
    for b in bodies[1:]:
        legs.setdefault(b.name.split("_")[0], []).append(b)
    ordered = [base]
    for leg in LEG_ORDER:
        hip, thigh, calf = legs[leg]
        assert hip.name.endswith("hip") and thigh.name.endswith("thigh") and calf.name.endswith("calf")
        ordered += [hip, thigh, calf]
    remap = {bodies.index(b): i for i, b in enumerate(ordered)}
    assert len(ordered) == N_BODIES_DYN
    reported = []
    for b in ordered:
        reported += b.reported
    assert len(reported) == N_BODIES_REPORTED, reported
    m = {
        "name": "go1",
        "body_names": [b.name for b in ordered],
        "reported_body_names": reported,
        "dof_names": [b.joint.name for b in ordered[1:]],
        "parent": [(-1 if b.parent < 0 else remap[b.parent]) for b in ordered],
        "mass": [b.mass for b in ordered],
        "com": [b.com.tolist() for b in ordered],
        "inertia": [b.inertia.tolist() for b in ordered],
        "joint_offset": [[0, 0, 0]] + [b.joint_t.tolist() for b in ordered[1:]],
        "joint_axis": [[0, 0, 0]] + [b.joint.axis.tolist() for b in ordered[1:]],
        "dof_lower": [b.joint.lower for b in ordered[1:]],
        "dof_upper": [b.joint.upper for b in ordered[1:]],
        "dof_effort": [b.joint.effort for b in ordered[1:]],
        "dof_velocity": [b.joint.velocity for b in ordered[1:]],
    }
    for b in ordered[1:]:
        assert np.allclose(b.joint_R, np.eye(3)), "engine assumes joint frames are pure translations (true for go1.urdf)"
    sph = _spheres_for_go1(ordered, reported)
    # priority order for the bounded contact list: feet first, then knees/legs, then trunk
    def prio(s):
        nm = reported[s[3]]
        return 0 if "foot" in nm else (1 if ("calf" in nm or "thigh" in nm) else (2 if "hip" in nm else 3))
    sph.sort(key=lambda s: (prio(s), s[0]))
    assert len(sph) <= MAX_SPHERES, len(sph)
    m["sphere_body"] = [int(s[0]) for s in sph]
    m["sphere_center"] = [np.asarray(s[1]).tolist() for s in sph]
    m["sphere_radius"] = [float(s[2]) for s in sph]
    m["sphere_reported"] = [int(s[3]) for s in sph]
    m["total_mass"] = float(sum(m["mass"]))
    return m


def build_object_model(urdf_path):
    """NPC objects: ball (free sphere), sheep (free upright cylinder -> 2 spheres), seesaw (fixed base + plank)."""
    links, joints, root = parse_urdf(urdf_path)
    bodies = collapse(links, joints, root)
    name = os.path.basename(urdf_path).split(".")[0]
    out = {"name": name, "bodies": []}
    for b in bodies:
        d = {"name": b.name, "mass": b.mass, "com": b.com.tolist(), "inertia": b.inertia.tolist(), "parent": b.parent,
             "shapes": [(k, (list(p) if not isinstance(p, float) else p), R.tolist(), t.tolist()) for k, p, R, t, _ in b.shapes]}
        if b.joint is not None:
            d.update(joint_offset=b.joint_t.tolist(), joint_axis=b.joint.axis.tolist(), lower=b.joint.lower,
                     upper=b.joint.upper, velocity=b.joint.velocity, effort=b.joint.effort)
        out["bodies"].append(d)
    return out


ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")


def load_model(name, resources_root=None):
    """Load <assets>/<name>_model.json, or build it from a URDF under an MQE `resources/` tree."""
    p = os.path.join(ASSET_DIR, f"{name}_model.json")
    if resources_root is None and os.path.isfile(p):
        with open(p) as f:
            return json.load(f)
    rel = {"go1": "robots/go1/urdf/go1.urdf", "ball": "objects/ball.urdf", "sheep": "objects/sheep.urdf",
           "seesaw": "objects/seesaw.urdf", "box": "objects/box.urdf", "rotation": "objects/rotation_door.urdf",
           "bridge": "objects/bridge/urdf/bridge.urdf", "circular": "objects/cylinder.urdf", "wrestling": "objects/wrestling_field/urdf/wrestling.urdf"}[name]
    path = os.path.join(resources_root, rel)
    return build_go1_model(path) if name == "go1" else build_object_model(path)


if __name__ == "__main__":  # regenerate the json assets from an MQE checkout:  python urdf_model.py <resources dir>
    import sys
    res = sys.argv[1]
    os.makedirs(ASSET_DIR, exist_ok=True)
    for nm in ("go1", "ball", "sheep", "seesaw", "box", "rotation", "bridge", "wrestling", "circular"):
        mdl = load_model(nm, res)
        with open(os.path.join(ASSET_DIR, f"{nm}_model.json"), "w") as f:
            json.dump(mdl, f, indent=1)
        print(nm, "ok")
