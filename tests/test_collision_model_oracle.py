"""The round-3 collision model in the ORACLE, against answers computed here from the model file alone (tests/rigid_ref.py: a float64
forward kinematics, nothing of the engine's geometry code): contacts found by the feature-point-vs-primitive tests have the
separation, normal and links that the URDF shapes dictate.  The HIP engine is held to the oracle's contact lists in
tests/test_gpu_parity.py; this file is what ties the oracle's lists to the geometry."""
import numpy as np
import pytest
import torch

import rigid_ref as rr
from helpers import make_desc, oracle_engine
from mqe.engine import abi
from mqe.utils import urdf_model

pytestmark = pytest.mark.usefixtures("solver")      # every test under both contact solvers (conftest.py)
STANCE = np.array([0.1, 0.8, -1.5, -0.1, 0.8, -1.5, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5])


def _scene(task, z=2.0, **kw):
    """robots frozen in the default stance high above the ground (no terrain contact), level, at rest"""
    d, k, _ = make_desc(task, 1, **kw)
    e = oracle_engine(d, k, f64=True)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    A = d.num_agents
    for a in range(A):
        dof[0, a * 12:(a + 1) * 12, 0] = torch.tensor(STANCE, dtype=torch.float32)
    dof[..., 1] = 0
    root[0, :, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    root[0, :, 7:] = 0
    root[0, :A, 2] = z
    root[0, 1, 1] = root[0, 0, 1] + 5.0          # the second robot out of the way
    return e, d, root, dof


def test_exact_model_collides_the_thigh_and_calf_boxes_themselves():
    """collision_model = "exact" (round 4): the thigh and calf bars are the URDF's boxes (go1.urdf:170,198: 213 x 24.5 x 34 mm and
    213 x 16 x 16 mm, link-aligned).  The ball beside a thigh's FLAT side face and beside its EDGE sees the box -- 12.25 mm from the
    axis on the face, hypot(12.25, 17) = 20.95 mm on the edge -- where the capsule model sees r = 16.6 mm all around."""
    m0 = urdf_model.load_model("go1")
    m = dict(m0, **m0["exact"])
    assert m["prim_type"][3] == 2 and m["prim_body"][3] == 2 and np.allclose(sorted(m["prim_half"][3]), [0.01225, 0.017, 0.1065])
    e, d, root, dof = _scene("go1football-1vs1", collision_model="exact")
    assert d.robot.n_spheres == 60 and d.robot.prim_type[3] == 2
    A, r = d.num_agents, d.npc_sphere_radius[0]
    base = root[0, 0, :3].numpy().astype(np.float64)
    c, _, Rl = _prim_world(m, 3, base)
    h = np.array(m["prim_half"][3])
    ay, ax = Rl[:, 1], Rl[:, 0]                     # the link's y axis (the bar's 24.5 mm) and x axis (34 mm)
    root[0, A, :3] = torch.tensor(c + ay * (h[1] + r + 0.005), dtype=torch.float32)           # beside the flat side face, 5 mm of air
    root[0, A, 7:] = 0
    _, _, con = e.debug_dynamics(0, 0)
    hit = [cc for cc in con if cc[1] == 2 and cc[2] == A]
    assert len(hit) == 1 and abs(hit[0][4] - 0.005) < 1e-6 and np.allclose(hit[0][5:8], -ay, atol=1e-5), con
    dg = (ay * h[1] + ax * h[0])
    n = (ay + ax) / np.sqrt(2.0)
    root[0, A, :3] = torch.tensor(c + dg + n * (r + 0.003), dtype=torch.float32)               # off the long edge, along its diagonal
    _, _, con = e.debug_dynamics(0, 0)
    hit = [cc for cc in con if cc[1] == 2 and cc[2] == A]
    assert len(hit) == 1 and abs(hit[0][4] - 0.003) < 1e-6 and np.allclose(hit[0][5:8], -n, atol=1e-5), con
    # the same ball positions against the capsule model: 16.6 mm all around -> 4.35 mm closer on the face, 4.35 mm farther on the edge
    e2, d2, root2, _ = _scene("go1football-1vs1")
    root2[0, A, :3] = torch.tensor(c + ay * (h[1] + r + 0.005), dtype=torch.float32)
    root2[0, A, 7:] = 0
    _, _, con2 = e2.debug_dynamics(0, 0)
    hit2 = [cc for cc in con2 if cc[1] == 2 and cc[2] == A]
    assert len(hit2) == 1 and abs(hit2[0][4] - (0.005 + h[1] - m0["prim_half"][3][0])) < 2e-4, con2


def _prim_world(m, q, base_p):
    Rb, pb = rr.fk(rr.load_model(), np.asarray(base_p, np.float64), np.eye(3), STANCE)
    b = m["prim_body"][q]
    return pb[b] + Rb[b] @ np.array(m["prim_center"][q]), Rb[b] @ np.array(m["prim_axis"][q]), Rb[b]


def test_ball_against_capsule_box_and_foot():
    """go1football-1vs1: the ball (sphere r) set beside a thigh capsule, on top of the trunk box and under a foot sphere of the floating
    robot: ONE contact each, between the ball and that link, with the gap / normal of sphere-vs-capsule, sphere-vs-box, sphere-vs-sphere"""
    m = urdf_model.load_model("go1")
    e, d, root, dof = _scene("go1football-1vs1")
    A, r = d.num_agents, d.npc_sphere_radius[0]
    base = root[0, 0, :3].numpy().astype(np.float64)
    # FL thigh capsule = primitive 3 (body 2): beside its middle, along +y, 7 mm of air between the surfaces
    c, u, _ = _prim_world(m, 3, base)
    assert m["prim_type"][3] == 1 and m["prim_body"][3] == 2
    rq = m["prim_half"][3][0]
    uh = u / np.linalg.norm(u)
    side = np.array([0.0, 1.0, 0.0]) - uh[1] * uh          # perpendicular to the (pitched and rolled) bar, towards +y
    side /= np.linalg.norm(side)
    root[0, A, :3] = torch.tensor(c + side * (rq + r + 0.007), dtype=torch.float32)
    root[0, A, 7:] = 0
    _, _, con = e.debug_dynamics(0, 0)
    assert len(con) == 1 and con[0, 0] == 0 and con[0, 1] == 2 and con[0, 2] == A, con
    assert abs(con[0, 4] - 0.007) < 1e-6 and np.allclose(con[0, 5:8], -side, atol=1e-5), con          # normal from the ball (B) to the robot (A)
    # beyond the capsule's end: the distance is to the END POINT of the segment, not to the infinite line
    end = c + u
    dirv = u / np.linalg.norm(u)
    root[0, A, :3] = torch.tensor(end + dirv * (rq + r + 0.004), dtype=torch.float32)
    _, _, con = e.debug_dynamics(0, 0)
    hit = [cc for cc in con if cc[1] == 2]
    assert len(hit) == 1 and abs(hit[0][4] - 0.004) < 1e-6 and np.allclose(hit[0][5:8], -dirv, atol=1e-5), con
    # on top of the trunk box (primitive 0): 3 mm above its top face, off-centre
    c0, _, R0 = _prim_world(m, 0, base)
    h0 = np.array(m["prim_half"][0])
    root[0, A, :3] = torch.tensor(c0 + np.array([0.1, 0.02, h0[2] + r + 0.003]), dtype=torch.float32)
    _, _, con = e.debug_dynamics(0, 0)
    assert len(con) == 1 and con[0, 1] == 0 and con[0, 2] == A and abs(con[0, 4] - 0.003) < 1e-6 and np.allclose(con[0, 5:8], [0, 0, -1], atol=1e-5), con
    # against the box's top front edge: closest point on the edge, normal along the diagonal
    root[0, A, :3] = torch.tensor(c0 + np.array([h0[0] + (r + 0.002) / np.sqrt(2), 0.0, h0[2] + (r + 0.002) / np.sqrt(2)]), dtype=torch.float32)
    _, _, con = e.debug_dynamics(0, 0)
    trunk = [cc for cc in con if cc[1] == 0]
    assert len(trunk) >= 1 and abs(trunk[0][4] - 0.002) < 1e-6 and np.allclose(trunk[0][5:8], [-np.sqrt(0.5), 0, -np.sqrt(0.5)], atol=1e-5), con
    # under the FL foot (primitive 5, a sphere on body 3): 6 mm below it; the calf's end cap (r 9.7 mm inside the 20 mm foot) stays clear
    cf, _, _ = _prim_world(m, 5, base)
    rf = m["prim_half"][5][0]
    root[0, A, :3] = torch.tensor(cf - np.array([0.0, 0.0, rf + r + 0.006]), dtype=torch.float32)
    _, _, con = e.debug_dynamics(0, 0)
    assert len(con) == 1 and con[0, 1] == 3 and abs(con[0, 4] - 0.006) < 1e-6 and np.allclose(con[0, 5:8], [0, 0, 1], atol=1e-5), con


def test_trunk_corners_carry_a_collapsed_robot(solver):
    """a limp robot (zero torques) collapses from its stance onto its belly: the base comes to rest at the trunk box's half height
    above the slab -- its four lower corners are feature points -- with the robot's weight on the ground and a good share of it on
    the base link itself (round 2's sphere set floated the trunk on three r = 57 mm spheres: same height, but no flat face)"""
    d, k, _ = make_desc("go1gate", 1)
    e = oracle_engine(d, k)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    root[0, :, 7:] = 0
    e.tensor(abi.T_TORQUES).zero_()
    for _ in range(450):
        e.simulate()
    pose0 = root[0, :, :7].clone()
    for _ in range(50):
        e.simulate()
    m = urdf_model.load_model("go1")
    hz = m["prim_half"][0][2]
    z = root[0, :, 2].numpy()
    assert np.all(np.abs(z - (d.ground_z + hz)) < 4e-3), (z, d.ground_z + hz)              # resting on the box's lower face (contact margin + ERP slack)
    # at rest.  The temporal solver's VELOCITY keeps a ripple on a body lying on eight contacts: gravity enters once per step, the
    # first sub-step lets the trunk sink by up to g dt (dt / 4) = 0.06 mm wherever its contacts have slack, the later ones ask those
    # 0.05-0.1 mm back within 1.25 ms each (0.1 m/s; 0.2 rad/s about the 9 cm wide belly), and four cold-started sweeps do not settle
    # that.  The POSE is at rest under both solvers.
    # (height, roll and pitch; along the ground the lying robot creeps at ~2 cm/s and yaws under either solver: four cold-started sweeps
    # do not converge the friction rows of eight contacts on one body)
    dp = (root[0, :, :7] - pose0).abs()
    assert dp[:, 2:5].max() < 5e-4 and dp.max() < 1e-2, dp
    assert root[0, :, 7:].abs().max() < (0.05 if solver == "pgs" else 0.3)
    cf = e.tensor(abi.T_CONTACT_FORCE)[0].reshape(2, 17, 3)
    total = cf[..., 2].sum(1).numpy()
    mg = sum(d.robot.mass[b] for b in range(13)) * 9.81
    assert np.all(np.abs(total - mg) < 0.08 * mg), (total, mg)
    assert np.all(cf[:, 0, 2].numpy() > 0.2 * mg), cf[:, 0, 2]                                # a good share of it on the base link itself


def test_two_robots_touch_with_their_primitives():
    """robot 1 moved until its FL hip capsule's outer end is 5 mm from robot 0's trunk box side: the contact pairs robot 1's hip link
    (a feature point) with robot 0's base (a primitive) at that gap, normal along the boxes' y axis"""
    m = urdf_model.load_model("go1")
    e, d, root, dof = _scene("go1gate")
    base0 = root[0, 0, :3].numpy().astype(np.float64)
    c0, _, _ = _prim_world(m, 0, base0)
    h0 = np.array(m["prim_half"][0])
    # robot 1's FL hip capsule (primitive 2, body 1): its -y end
    cq, u, _ = _prim_world(m, 2, np.zeros(3))
    rq = m["prim_half"][2][0]
    end = cq + u if (cq + u)[1] < (cq - u)[1] else cq - u
    # place robot 1 on the +y side of robot 0, same height and heading, hip end facing the trunk's +y face, clear of robot 0's own legs in x
    target = c0 + np.array([0.0, h0[1] + rq + 0.005, 0.03])
    root[0, 1, :3] = torch.tensor(target - end, dtype=torch.float32)
    _, _, con = e.debug_dynamics(0, 0)
    pair = [cc for cc in con if cc[2] >= 0 and cc[0] != cc[2]]
    hip_trunk = [cc for cc in pair if {(int(cc[0]), int(cc[1])), (int(cc[2]), int(cc[3]))} == {(1, 1), (0, 0)}]
    assert hip_trunk, con
    assert any(abs(cc[4] - 0.005) < 2e-6 and abs(abs(cc[6]) - 1.0) < 1e-5 for cc in hip_trunk), hip_trunk


def test_box_corner_presses_into_the_trunk_face():
    """go1pushbox (round 4): the free box's own corners are tested against the robots' primitives.  A box balanced on one corner 3 mm above
    the middle of the trunk's top face -- no feature point of the robot (corners, capsule ends, feet) is anywhere near it -- gives ONE
    contact between the box and the base link, separation 3 mm, normal straight down onto the robot (from B, the box, to A)."""
    e, d, root, dof = _scene("go1pushbox")
    m = urdf_model.load_model("go1")
    A = d.num_agents
    base = root[0, 0, :3].numpy().astype(np.float64)
    c0, _, _ = _prim_world(m, 0, base)
    top = c0 + np.array([0.03, 0.01, m["prim_half"][0][2]])                  # a point on the trunk's top face, away from its edges
    h = np.array([d.npc_box_half[0], d.npc_box_half[1], d.npc_box_half[2]], np.float64)
    # rotation that turns the box's (-,-,-) diagonal straight down
    v = -h / np.linalg.norm(h); t = np.array([0.0, 0.0, -1.0])
    ax = np.cross(v, t); s_, c_ = np.linalg.norm(ax), float(v @ t); ax /= s_
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + s_ * K + (1 - c_) * K @ K
    ang = np.arctan2(s_, c_)
    quat = np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]])
    centre = top + np.array([0, 0, 0.003]) - R @ (-h)
    root[0, A, :3] = torch.tensor(centre, dtype=torch.float32)
    root[0, A, 3:7] = torch.tensor(quat, dtype=torch.float32)
    root[0, A, 7:] = 0
    _, _, con = e.debug_dynamics(0, 0)
    hit = [cc for cc in con if cc[2] == A]
    assert len(hit) == 1 and hit[0][0] == 0 and hit[0][1] == 0, con
    assert abs(hit[0][4] - 0.003) < 2e-6 and np.allclose(hit[0][5:8], [0, 0, -1], atol=1e-5), hit


def test_wall_corner_presses_into_the_side_of_the_trunk():
    """edge contacts (round 4, include/mqe_hip.h edge_contacts bit 1): a gate post's vertical corner 3 mm from the middle of the trunk's side
    face -- between the legs, no feature point of the robot near it, the wall's signed-distance field says 5 cm of air at every trunk
    corner -- gives ONE contact on the base link: separation 3 mm, normal from the post to the robot.  With edge_contacts = 0 (round 3)
    there is none."""
    m = urdf_model.load_model("go1")
    for mask, want in ((7, 1), (0, 0)):
        d, k, ctx = make_desc("go1gate", 1, edge_contacts=mask)
        t = ctx["terrain"]
        corners = np.unique(t.wall_corner.reshape(-1, 2), axis=0)
        gate = corners[np.argsort(np.abs(corners[:, 0] - corners[:, 0].mean()))[:4]]            # the four corners of the two gate posts (mid-track)
        lo = gate[gate[:, 1] < gate[:, 1].mean()]
        cx, cy = lo[np.argmin(lo[:, 0])]                                                          # the lower post's corner that faces the opening, near side
        e = oracle_engine(d, k, f64=True)
        e.reset_all()
        root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
        dof[0, :, 0] = torch.tensor(np.tile(STANCE, 2), dtype=torch.float32); dof[..., 1] = 0
        root[0, :, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0]); root[0, :, 7:] = 0
        hy = m["prim_half"][0][1]
        root[0, 0, 0] = float(cx) - m["prim_center"][0][0]                                               # the corner at the middle of the trunk box
        root[0, 0, 1] = float(cy) + hy + 0.003 - m["prim_center"][0][1]
        root[0, 0, 2] = 0.32
        root[0, 1, :2] = torch.tensor([1.5, 2.7])                                                 # the other robot out of the way
        _, _, con = e.debug_dynamics(0, 0)
        hit = [c for c in con if c[0] == 0 and c[1] == 0]
        assert len(hit) == want, (mask, con)
        if want:
            assert abs(hit[0][4] - 0.003) < 1e-5 and np.allclose(hit[0][5:8], [0, 1, 0], atol=1e-4), hit


def test_box_edge_cuts_into_a_thigh_and_into_the_trunk_face():
    """edge contacts bits 2 and 4 on go1pushbox: (a) the free box's top edge under the MIDDLE of a thigh capsule -- both ends of the capsule
    (its feature points) are centimetres away -- is found by the closest approach of the capsule's whole axis; (b) the box's vertical edge
    3 mm from the middle of the trunk's side face (box corners at z = 0 and 1 m, trunk corners 19 cm away) is found by the edge-vs-box
    search.  Expected separations from the model file's geometry alone."""
    m = urdf_model.load_model("go1")
    e, d, root, dof = _scene("go1pushbox", edge_contacts=7)         # (bit 4 -- box edges against box primitives -- exists in the oracle only)
    A = d.num_agents
    h = np.array([d.npc_box_half[0], d.npc_box_half[1], d.npc_box_half[2]], np.float64)
    base = root[0, 0, :3].numpy().astype(np.float64)
    # (a) FL thigh capsule (primitive 3): the box's vertical edge stands r + 4 mm off the bar's axis, on its outer side, BETWEEN the bar's
    # feature points (upper end, middle, knee)
    c, u, _ = _prim_world(m, 3, base)
    rq = m["prim_half"][3][0]
    uh = u / np.linalg.norm(u)
    nrm = np.cross(uh, [0.0, 0.0, 1.0]); nrm /= np.linalg.norm(nrm)       # the common perpendicular of the bar and a vertical line
    if nrm[1] > 0:
        nrm = -nrm                                         # from the edge (outside, +y of the left leg) to the bar
    line = (c - 0.4 * u) - nrm * (rq + 0.004)              # at 30 % of the axis: 4 cm from the bar's middle (a feature point), 6 cm from its upper end
    root[0, A, :3] = torch.tensor([line[0] + h[0], line[1] + h[1], float(base[2])], dtype=torch.float32)      # the box's (-x, -y) vertical edge on that line
    root[0, A, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0]); root[0, A, 7:] = 0
    _, _, con = e.debug_dynamics(0, 0)
    hit = [cc for cc in con if cc[0] == 0 and cc[1] == 2 and cc[2] == A]
    # (the golden-section search stops within 0.05 % of the segment: the separation is flat there, the normal within a milliradian)
    assert len(hit) == 1 and abs(hit[0][4] - 0.004) < 2e-5 and np.allclose(hit[0][5:8], nrm, atol=2e-3), (con, nrm)
    # (b) the box's vertical edge (x = -hx, y = +hy ... chosen by position) against the trunk's +y side face, mid-length
    c0, _, _ = _prim_world(m, 0, base)
    hy0 = m["prim_half"][0][1]
    line = np.array([c0[0], c0[1] + hy0 + 0.003])          # where the vertical edge stands (x, y)
    # the box turned by 45 degrees about z: its (-x, -y) vertical edge points at the trunk like a wedge, the faces recede at 45 degrees
    root[0, A, :3] = torch.tensor([line[0], line[1] + h[0] * np.sqrt(2.0), float(base[2])], dtype=torch.float32)
    root[0, A, 3:7] = torch.tensor([0.0, 0.0, np.sin(np.pi / 8), np.cos(np.pi / 8)], dtype=torch.float32)
    _, _, con = e.debug_dynamics(0, 0)
    hit = [cc for cc in con if cc[0] == 0 and cc[1] == 0 and cc[2] == A]
    assert len(hit) == 1 and abs(hit[0][4] - 0.003) < 2e-5 and np.allclose(hit[0][5:8], [0, -1, 0], atol=1e-3), con
