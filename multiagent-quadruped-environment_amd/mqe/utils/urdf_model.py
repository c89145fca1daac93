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

Collision geometry of the Go1: the URDF's own primitives (trunk and head boxes, hip capsules, foot spheres; the thigh and
calf bars as capsules) plus their feature points -- `_collision_model_for_go1`, DESIGN.md "collision model".
"""
import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

LEG_ORDER = ("FL", "FR", "RL", "RR")
MAX_SPHERES = 64
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


PRIM_SPHERE, PRIM_CAPSULE, PRIM_BOX = 0, 1, 2
MAX_PRIMS = 20
SELF_PAIR_MARGIN = 0.03       # a (feature, primitive) pair is a self-collision candidate if some pose brings it this close [m]
SELF_SAFE_MARGIN = 0.04       # inside the "safe box" of joint angles every candidate pair stays farther apart than this [m]
# the stance the safe box is grown around: default_joint_angles of go1_config.py:88-103 in this model's joint order (FL, FR, RL, RR)
SELF_SAFE_STANCE = [0.1, 0.8, -1.5, -0.1, 0.8, -1.5, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5]


def _capsule_for_bar(h):
    """Sphere-swept segment standing in for a slender URDF box (thigh 213 x 24.5 x 34 mm, calf 213 x 16 x 16 mm; go1.urdf thigh / calf
    <collision>): the radius that minimises the two-sided deviation of the cross-section (half the sum of the in-circle and the
    circum-circle radii of the rectangle) and the half-length at which the cap's overshoot along the bar equals its shortfall at the
    end corners.  Returns (long axis index, half-length of the segment, radius)."""
    h = np.asarray(h, np.float64)
    order = np.argsort(h)
    long_ax = int(order[2])
    rin, rout = float(h[order[0]]), float(np.hypot(h[order[0]], h[order[1]]))
    r = 0.5 * (rin + rout)
    H = float(h[long_ax])
    lo, hi = max(H - 2 * r, 0.0), H
    for _ in range(60):                      # bisection on a: overshoot(a) = a + r - H rises, corner shortfall falls
        a = 0.5 * (lo + hi)
        if a + r - H > np.hypot(H - a, rout) - r:
            hi = a
        else:
            lo = a
    return long_ax, 0.5 * (lo + hi), r


def _collision_model_for_go1(bodies, reported_names, exact=False):
    """The Go1 collision model of the engine: the URDF's 18 primitives themselves (go1.urdf:56 trunk box, :80 head box, hip cylinders
    -> capsules as `replace_cylinder_with_capsule` does (go1_config.py:75), thigh / calf bars -> capsules (`_capsule_for_bar`), foot
    spheres) = what OTHER bodies collide with, and their FEATURE POINTS (sphere-swept: capsule end points with the capsule's radius,
    box corners with radius 0, the foot spheres) = what is tested against the terrain maps, the scenery and the other actors'
    primitives.  A convex body's lowest / outermost point against a plane is always a feature point, so ground and wall-face
    contacts are those of the primitives themselves.
    Returns (prims, feats): prims = [dict(type, body, reported, center, axis (capsule half-segment), half (box) | radius)],
    feats = [dict(body, reported, center, radius, prim)], both in the link frame of `body`."""
    prims, feats = [], []
    for bi, b in enumerate(bodies):
        for kind, prm, R, t, rep in b.shapes:
            ri = reported_names.index(rep)
            R, t = np.asarray(R, np.float64), np.asarray(t, np.float64)
            pi = len(prims)
            if kind == "sphere":
                prims.append(dict(type=PRIM_SPHERE, body=bi, reported=ri, center=t, axis=np.zeros(3), half=np.array([prm, 0, 0.0]), bound=float(prm)))
                feats.append(dict(body=bi, reported=ri, center=t, radius=float(prm), prim=pi, tag="foot"))
            elif kind == "cylinder":        # capsule of the cylinder's radius whose segment is the cylinder's axis (Isaac Gym's replacement)
                r, L = prm
                u = R @ np.array([0.0, 0.0, L / 2])
                prims.append(dict(type=PRIM_CAPSULE, body=bi, reported=ri, center=t, axis=u, half=np.array([r, 0, 0.0]), bound=float(r + L / 2)))
                for sgn in (-1.0, 1.0):
                    feats.append(dict(body=bi, reported=ri, center=t + sgn * u, radius=float(r), prim=pi, tag="hip"))
            elif kind == "box" and b.parent < 0:      # trunk / head: the box itself, corners as feature points
                h = np.asarray(prm, np.float64)
                assert np.allclose(np.abs(R), np.round(np.abs(R)), atol=1e-9), "base boxes must be aligned with the base frame"
                hl = np.abs(R) @ h                     # half extents along the link axes
                la = int(np.argmax(hl))
                ax = np.zeros(3); ax[la] = hl[la]          # a box's "axis": half of its longest edge (its bounding capsule, for screens)
                prims.append(dict(type=PRIM_BOX, body=bi, reported=ri, center=t, axis=ax, half=hl, bound=float(np.linalg.norm(hl))))
                corners = [np.array([sx, sy, sz]) * hl for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
                is_head = len([p for p in prims if p["type"] == PRIM_BOX]) > 1
                for c in corners:
                    # the head box (4 x 10 x 10 cm in front of the trunk): its four rear corners are never the outermost point of
                    # the trunk + head pair by more than 3 mm (y: 50 vs 46.75 mm, z: 50 vs 57 mm), so only the front four are kept
                    # (the exact model keeps all eight)
                    if is_head and c[0] < 0 and not exact:
                        continue
                    feats.append(dict(body=bi, reported=ri, center=t + c, radius=0.0, prim=pi, tag="head" if is_head else "trunk"))
            elif kind == "box" and exact:   # thigh / calf bar as the URDF has it: a link-aligned box (go1.urdf:170,198: rpy = (0, pi / 2, 0))
                h = np.asarray(prm, np.float64)
                assert np.allclose(np.abs(R), np.round(np.abs(R)), atol=1e-9), "leg bars must be aligned with their link frame"
                hl = np.abs(R) @ h
                la = int(np.argmax(hl))
                ax = np.zeros(3); ax[la] = hl[la]
                prims.append(dict(type=PRIM_BOX, body=bi, reported=ri, center=t, axis=ax, half=hl, bound=float(np.linalg.norm(hl))))
                far = 1.0 if t[la] > 0 else -1.0         # the end of the bar away from the link's own joint (the joint sits at the link origin)
                for sa in (-1, 1):
                    for sb in (-1, 1):
                        c = np.zeros(3)
                        c[la] = (far if "thigh" in b.name else -far) * hl[la]
                        o = [k for k in range(3) if k != la]
                        c[o[0]], c[o[1]] = sa * hl[o[0]], sb * hl[o[1]]
                        # thigh: the four corners of its LOWER end (the knee); its upper corners lie 14 mm inside the hip capsule whatever the
                        # joint angles.  calf: the four corners of its UPPER end (they stick out of the folded knee by up to 8 mm sin(angle));
                        # its lower corners lie inside the foot sphere (11.3 mm from its centre, r = 20 mm).
                        feats.append(dict(body=bi, reported=ri, center=t + c, radius=0.0, prim=pi, tag="knee" if "thigh" in b.name else "thigh"))
            elif kind == "box":             # thigh / calf bar -> capsule
                h = np.asarray(prm, np.float64)
                long_ax, a, r = _capsule_for_bar(h)
                e = np.zeros(3); e[long_ax] = 1.0
                u = R @ (a * e)
                if np.linalg.norm(t + u) < np.linalg.norm(t - u):   # +u points away from the link's own joint (towards the child)
                    u = -u
                prims.append(dict(type=PRIM_CAPSULE, body=bi, reported=ri, center=t, axis=u, half=np.array([r, 0, 0.0]), bound=float(r + a)))
                if "calf" in b.name:
                    # the calf's lower end lies inside the foot sphere (r 20 mm) and its upper end inside the thigh's lower end cap
                    # (same point -- the knee -- and the thigh is the thicker bar): the calf contributes the primitive only
                    continue
                # lower end = the knee; the upper end lies inside the hip capsule (17 mm from its segment, 46 mm radius), so the thigh's
                # second feature point is its MIDDLE: what a wall's or a box's edge meets when it cuts into the bar between its ends
                feats.append(dict(body=bi, reported=ri, center=t + u, radius=float(r), prim=pi, tag="knee"))
                feats.append(dict(body=bi, reported=ri, center=t, radius=float(r), prim=pi, tag="thigh"))
    return prims, feats


def _self_pair_candidates(m, prims, feats, n_samples=200000, seed=0):
    """(feature, primitive) pairs of one robot that can come within SELF_PAIR_MARGIN of each other somewhere inside the joint limits
    (links neither the same nor parent and child: PhysX filters exactly the adjacent links of an articulation).  Sampled: uniform
    joint angles, distance = feature sphere against the primitive's bounding capsule / box (exact for spheres and capsules)."""
    rng = np.random.RandomState(seed)
    par = m["parent"]
    lo, hi = np.asarray(m["dof_lower"]), np.asarray(m["dof_upper"])
    off, axs = np.asarray(m["joint_offset"], np.float64), np.asarray(m["joint_axis"], np.float64)
    # (a sphere against a sphere is the same pair from either side: kept once, lower primitive first)
    cand = [(i, j) for i, f in enumerate(feats) for j, q in enumerate(prims)
            if f["prim"] != j and f["body"] != q["body"] and par[f["body"]] != q["body"] and par[q["body"]] != f["body"]
            and not (q["type"] == PRIM_SPHERE and prims[f["prim"]]["type"] == PRIM_SPHERE and f["prim"] > j)]
    best = np.full(len(cand), np.inf)

    def gaps(q):
        """(B, n_cand) signed gaps of every candidate pair at joint angles q (B, 12)"""
        B = len(q)
        R = np.zeros((B, N_BODIES_DYN, 3, 3)); R[:, 0] = np.eye(3)
        p = np.zeros((B, N_BODIES_DYN, 3))
        for b in range(1, N_BODIES_DYN):
            ax = axs[b]
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            ang = q[:, b - 1]
            Rj = np.eye(3)[None] + np.sin(ang)[:, None, None] * K[None] + (1 - np.cos(ang))[:, None, None] * (K @ K)[None]
            R[:, b] = R[:, par[b]] @ Rj
            p[:, b] = p[:, par[b]] + R[:, par[b]] @ off[b]
        fc = np.stack([p[:, f["body"]] + R[:, f["body"]] @ f["center"] for f in feats], 1)          # (B, F, 3)
        qc = np.stack([p[:, g["body"]] + R[:, g["body"]] @ g["center"] for g in prims], 1)
        qu = np.stack([R[:, g["body"]] @ g["axis"] for g in prims], 1)
        out = np.empty((B, len(cand)))
        for k, (i, j) in enumerate(cand):
            g = prims[j]
            d = fc[:, i] - qc[:, j]
            if g["type"] == PRIM_BOX:        # (the exact box, not its bounding capsule)
                loc = np.einsum("bji,bj->bi", R[:, g["body"]], d)
                ex = np.maximum(np.abs(loc) - g["half"], 0.0)
                dist = np.linalg.norm(ex, axis=1)
            else:
                uu = max(float(g["axis"] @ g["axis"]), 1e-18)
                tpar = np.clip(np.einsum("bi,bi->b", d, qu[:, j]) / uu, -1, 1)
                dist = np.linalg.norm(d - tpar[:, None] * qu[:, j], axis=1) - g["half"][0]
            out[:, k] = dist - feats[i]["radius"]
        return out
    B = 2000
    for _ in range(n_samples // B):
        best = np.minimum(best, gaps(lo + (hi - lo) * rng.rand(B, 12)).min(0))
    keep = [c for c, b in zip(cand, best) if b < SELF_PAIR_MARGIN]
    # A joint-space box around the stance inside which NO candidate pair comes within SELF_SAFE_MARGIN: an env whose joints are all
    # inside it skips the self-collision phase with one ballot (a walking robot is).  The box is the stance widened per joint type
    # (hip / thigh / calf, the same for all legs) by the largest scale at which a dense sample INSIDE the box finds nothing closer.
    stance = np.asarray(SELF_SAFE_STANCE, np.float64)
    wid_lo, wid_hi = np.array([0.5, 0.9, 1.1] * 4), np.array([0.5, 0.9, 0.55] * 4)
    blo, bhi = stance.copy(), stance.copy()
    for scale in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2):
        tlo, thi = np.maximum(stance - scale * wid_lo, lo), np.minimum(stance + scale * wid_hi, hi)
        worst = np.inf
        for _ in range(150):
            worst = min(worst, float(gaps(tlo + (thi - tlo) * rng.rand(B, 12)).min()))
            if worst < SELF_SAFE_MARGIN:
                break
        if worst >= SELF_SAFE_MARGIN:
            blo, bhi = tlo, thi
            break
    return keep, len(cand), blo, bhi


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
    def pack(out, exact):
        prims, feats = _collision_model_for_go1(ordered, reported, exact=exact)
        # priority order of the feature points for the bounded contact list: feet first (they carry the robot), then the trunk and head
        # corners (base contact is what check_termination thresholds), then knees, thigh tops and hips
        rank = {"foot": 0, "trunk": 1, "head": 2, "knee": 3, "thigh": 4, "hip": 5}
        feats.sort(key=lambda f: (rank[f["tag"]], f["body"]))
        assert len(feats) <= MAX_SPHERES and len(prims) <= MAX_PRIMS, (len(feats), len(prims))
        out["sphere_body"] = [int(f["body"]) for f in feats]
        out["sphere_center"] = [np.asarray(f["center"]).tolist() for f in feats]
        out["sphere_radius"] = [float(f["radius"]) for f in feats]
        out["sphere_reported"] = [int(f["reported"]) for f in feats]
        out["sphere_prim"] = [int(f["prim"]) for f in feats]
        out["sphere_tag"] = [f["tag"] for f in feats]
        out["prim_type"] = [int(g["type"]) for g in prims]
        out["prim_body"] = [int(g["body"]) for g in prims]
        out["prim_reported"] = [int(g["reported"]) for g in prims]
        out["prim_center"] = [np.asarray(g["center"]).tolist() for g in prims]
        out["prim_axis"] = [np.asarray(g["axis"]).tolist() for g in prims]
        out["prim_half"] = [np.asarray(g["half"]).tolist() for g in prims]
        out["prim_bound"] = [float(g["bound"]) for g in prims]
        # upper bound of |feature point - base origin| + its radius over all joint angles: joint offsets along the chain + the local centre

        def chain(b):
            return 0.0 if b == 0 else float(np.linalg.norm(m["joint_offset"][b])) + chain(m["parent"][b])
        out["feature_reach"] = max(chain(f["body"]) + float(np.linalg.norm(f["center"])) + f["radius"] for f in feats)
        pairs, n_all, safe_lo, safe_hi = _self_pair_candidates(m, prims, feats)
        out["self_pairs"] = [[int(i), int(j)] for i, j in pairs]
        out["self_pairs_unpruned"] = int(n_all)
        out["self_safe_lo"], out["self_safe_hi"] = [float(x) for x in safe_lo], [float(x) for x in safe_hi]
    pack(m, False)
    # the same robot with the thigh and calf bars as the URDF's own boxes (go1.urdf:170,198) and every corner of theirs that can be
    # outermost as a feature point: desc `collision_model = "exact"` (mqe/engine/desc.py)
    m["exact"] = {}
    pack(m["exact"], True)
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
