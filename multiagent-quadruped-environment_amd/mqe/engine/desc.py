"""MQE config class -> `mqe_sim_desc` (include/mqe_hip.h).  Everything the reference hands to Isaac Gym at scene
construction (legged_robot.py:754-923, go1.py:357-479, legged_robot_config.py:211-229) ends up in this one struct."""
import ctypes as C
import math
import os

import numpy as np

from . import abi
from ..utils import policy_weights, urdf_model

_ANNOUNCED = {}        # environment overrides already announced on stderr

# reward-term order per task: (scale attribute on cfg.rewards.scales, key in the wrapper's reward_buffer)
REWARD_TERMS = {
    "gate": [("target_reward_scale", "target reward"), ("contact_punishment_scale", "contact punishment"),
             ("success_reward_scale", "success reward"), ("agent_distance_punishment_scale", "agent distance punishment")],
    "sheep": [("success_reward_scale", "success reward"), ("contact_punishment_scale", "contact punishment"),
              ("sheep_movement_reward_scale", "sheep movement reward"), ("mixed_sheep_reward_scale", "mixed sheep reward"),
              ("sheep_pos_var_exp_punishment_scale", "sheep pos var punishment"), ("sheep_pos_var_lin_punishment_scale", None)],
    "seesaw": [("x_movement_reward_scale", "x movement reward"), ("height_reward_scale", "height reward"),
               ("y_punishment_scale", "y punishment"), ("contact_punishment_scale", "contact punishment"),
               ("agent_distance_punishment_scale", "agent distance punishment"), ("success_reward_scale", "success reward"),
               ("fall_punishment_scale", "fall punishment")],
    "football_defender": [("goal_reward_scale", "goal reward"), ("ball_gate_distance_reward_scale", "ball gate distance reward")],
    # the wrapper overwrites the configured scale with 1 after reading it (go1_pushbox_wrapper.py:20), see build_desc
    "pushbox": [("box_x_movement_reward_scale", "box movement reward")],
    # hard-set in the wrapper's constructor (go1_rotation_wrapper.py:18-20): 5 / 1 / 1, see build_desc
    "rotation": [("success_reward_scale", "success reward"), ("punishment_scale", "punishment"), ("distance_reward_scale", "distance reward")],
    "bridge": [("success_reward_scale", "success reward"), ("punishment_scale", "punishment"), ("target_reward_scale", "target reward")],
    "wrestling": [("success_reward_scale", "success reward"), ("punishment_scale", "punishment")],
    # go1_tug_wrapper.py:21-33: four reward terms and six running logs of positions share the reward_buffer
    "tug": [("success_reward_scale", "success reward"), ("punishment_reward_scale", "punishment"), ("pos_reward_scale", "pos reward"),
            ("pos_punishment_scale", "pos punishment"), (None, "npc pos"), (None, "total reward"), (None, "pos"), (None, "pos_y"),
            (None, "opponet pos"), (None, "opponet pos_y")],
    "plain": [],
}


def fill_command_obs(cfg):
    """Default 70-float locomotion observation and the action-slot layout (reference go1.py:411-479)."""
    dc, sc, cc = cfg.control.default_command, cfg.control.obs_scales, cfg.command.cfg
    o = np.zeros(70, np.float32)
    idx = {}
    n = 0
    if not cc.vel:
        o[3], o[4], o[5] = dc.lin_vel_x * sc.lin_vel, dc.lin_vel_y * sc.lin_vel, dc.ang_vel * sc.ang_vel
    else:
        idx["vel"], n = n, n + 3
    for flag, slot, val, scale in (("body_height", 6, dc.body_height, sc.body_height), ("gait_freq", 7, dc.gait_freq, sc.gait_freq)):
        if not getattr(cc, flag):
            o[slot] = val * scale
        else:
            idx[flag], n = n, n + 1
    if not cc.gait:
        g = cfg.command.gaits[dc.gait]
        o[8], o[9], o[10], o[11] = g[0] * sc.gait_phase, g[1] * sc.gait_phase, g[2] * sc.gait_phase, 0.5 * sc.gait_phase
    else:
        idx["gait"], n = n, n + 4
    if not cc.footswing_height:
        o[12] = dc.footswing_height * sc.footswing_height
    else:
        idx["footswing_height"], n = n, n + 1
    if not cc.body_pose:
        o[13], o[14] = dc.body_pitch * sc.body_pitch, dc.body_roll * sc.body_roll
    else:
        idx["body_pose"], n = n, n + 2
    for flag, slot, val, scale in (("stance_width", 15, dc.stance_width, sc.stance_width),
                                   ("stance_length", 16, dc.stance_length, sc.stance_length),
                                   ("aux_reward", 17, dc.aux_reward, sc.aux_reward)):
        if not getattr(cc, flag):
            o[slot] = val * scale
        else:
            idx[flag], n = n, n + 1
    return o, idx


def _fp(arr, keep):
    a = np.ascontiguousarray(arr, dtype=np.float32)
    keep.append(a)
    return a.ctypes.data_as(abi.FP)


def _fill_mlp(m, Ws, bs, keep):
    m.n_layers = len(Ws)
    m.dims[0] = Ws[0].shape[1]
    for i, (w, b) in enumerate(zip(Ws, bs)):
        m.dims[i + 1] = w.shape[0]
        m.W[i] = _fp(w, keep)
        m.b[i] = _fp(b, keep)


def task_kind(cfg):
    name = getattr(cfg.env, "env_name", "")
    npc = getattr(cfg.asset, "name_npc", "")
    if name == "go1gate":
        return "gate"
    if name == "go1sheep":
        return "sheep"
    if name == "go1seesaw":
        return "seesaw"
    if name == "go1pushbox":
        return "pushbox"
    if name == "go1rotationCfg":
        return "rotation"
    if name == "go1tug":
        return "tug"
    if name == "go1bridge":
        return "bridge"
    if name == "go1wrestling":
        return "wrestling"
    if name == "go1football" and cfg.env.num_agents == 3 and npc == "ball":
        return "football_defender"
    return "plain"


def build_desc(cfg, num_envs, terrain, env_origins, agent_origins, gate_pos=None, env_id_offset=0, seed=0,
               task=None, body=None, resources_root=None, solver_iterations=None, erp=0.2, noise_mode=0, solver_type=None, velocity_iterations=None,
               terrain_levels=None, terrain_types=None, collision_model=None, edge_contacts=None):
    """Returns (SimDesc, keepalive) -- keepalive holds the numpy arrays the struct points into."""
    keep = []
    d = abi.SimDesc()
    A = getattr(cfg.env, "num_agents", 1)
    P = getattr(cfg.env, "num_npcs", 0)
    task = task or task_kind(cfg)
    d.abi_version = abi.ABI_VERSION
    d.num_envs, d.num_agents, d.num_npcs = num_envs, A, P
    d.npc_kind = abi.NPC[getattr(cfg.asset, "name_npc", "") or "none"]
    d.task = abi.TASK[task]
    d.env_id_offset, d.seed = env_id_offset, seed
    px = cfg.sim.physx
    d.dt, d.decimation, d.gravity_z = cfg.sim.dt, cfg.control.decimation, cfg.sim.gravity[2]
    # PhysX TGS position iterations (legged_robot_config.py:221) -> projected Gauss-Seidel sweeps of the engine
    d.solver_iterations = int(solver_iterations if solver_iterations is not None else px.num_position_iterations)
    d.contact_offset, d.max_depenetration_velocity = px.contact_offset, px.max_depenetration_velocity
    d.friction = 0.5 * (cfg.terrain.static_friction + 1.0)   # average of terrain and (default 1.0) shape friction
    d.erp = erp
    # sim.physx.solver_type (legged_robot_config.py:219: 0 pgs, 1 tgs) picks the contact solver of the engine (include/mqe_hip.h);
    # MQE_SOLVER=pgs|tgs overrides it for A/B runs (announced)
    d.solver_type = int(solver_type if solver_type is not None else getattr(px, "solver_type", 1))
    if solver_type is None and os.environ.get("MQE_SOLVER"):
        import sys
        d.solver_type = {"pgs": 0, "tgs": 1}[os.environ["MQE_SOLVER"]]
        if _ANNOUNCED.setdefault("MQE_SOLVER") != os.environ["MQE_SOLVER"]:
            _ANNOUNCED["MQE_SOLVER"] = os.environ["MQE_SOLVER"]
            print(f"mqe: override in effect: MQE_SOLVER={os.environ['MQE_SOLVER']}", file=sys.stderr)
    d.velocity_iterations = int(velocity_iterations if velocity_iterations is not None else getattr(px, "num_velocity_iterations", 0))
    # robot model
    m = urdf_model.load_model("go1", resources_root)
    # collision model of the robot (include/mqe_hip.h mqe_robot_model): "capsule" (default) or "exact" (thigh / calf as the URDF's boxes,
    # 60 feature points); cfg.asset.collision_model or MQE_COLLISION_MODEL pick it when the caller does not
    cm = collision_model or os.environ.get("MQE_COLLISION_MODEL") or getattr(cfg.asset, "collision_model", "capsule")
    assert cm in ("capsule", "exact"), cm
    if cm == "exact":
        m = dict(m, **m["exact"])
    r = d.robot
    for b in range(abi.NBODY):
        r.mass[b] = m["mass"][b]
        I = np.asarray(m["inertia"][b])
        for k, v in enumerate((I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2])):
            r.inertia[b][k] = v
        for k in range(3):
            r.com[b][k] = m["com"][b][k]
            r.joint_offset[b][k] = m["joint_offset"][b][k]
            r.joint_axis[b][k] = m["joint_axis"][b][k]
    for j in range(abi.NDOF):
        r.dof_lower[j], r.dof_upper[j] = m["dof_lower"][j], m["dof_upper"][j]
        r.dof_vel_limit[j] = m["dof_velocity"][j]          # props["velocity"] (legged_robot.py:315): enforced by the solver
    # collision model: feature points + the URDF's primitives (mqe/utils/urdf_model.py::_collision_model_for_go1)
    r.n_spheres = len(m["sphere_body"])
    for s in range(r.n_spheres):
        r.sphere_body[s], r.sphere_reported[s], r.sphere_radius[s] = m["sphere_body"][s], m["sphere_reported"][s], m["sphere_radius"][s]
        r.sphere_prim[s] = m["sphere_prim"][s]
        for k in range(3):
            r.sphere_center[s][k] = m["sphere_center"][s][k]
    r.feature_reach = m["feature_reach"]
    r.n_prims = len(m["prim_type"])
    for q in range(r.n_prims):
        r.prim_type[q], r.prim_body[q], r.prim_reported[q], r.prim_bound[q] = m["prim_type"][q], m["prim_body"][q], m["prim_reported"][q], m["prim_bound"][q]
        for k in range(3):
            r.prim_center[q][k], r.prim_axis[q][k], r.prim_half[q][k] = m["prim_center"][q][k], m["prim_axis"][q][k], m["prim_half"][q][k]
    # self-collision (asset.self_collisions is Isaac Gym's filter mask: 0 = links of one robot collide, go1_config.py:73):
    # the model file's (feature point, primitive) candidates -- links neither the same nor adjacent, reachable inside the joint limits
    for j in range(abi.NDOF):
        r.self_safe_lo[j], r.self_safe_hi[j] = m["self_safe_lo"][j], m["self_safe_hi"][j]
    pairs = [tuple(p) for p in m["self_pairs"]]
    assert len(pairs) <= abi.MAX_SELF_PAIRS, len(pairs)
    r.n_self_pairs = len(pairs)
    for k, (i, j) in enumerate(pairs):
        r.self_pair[k] = i | (j << 8)
    d.self_collision = 1 if int(getattr(cfg.asset, "self_collisions", 1)) == 0 else 0
    # NPC objects
    if d.npc_kind in (abi.NPC["ball"], abi.NPC["sheep"]):
        om = urdf_model.load_model("ball" if d.npc_kind == abi.NPC["ball"] else "sheep", resources_root)["bodies"][0]
        d.npc_mass, d.npc_inertia = om["mass"], om["inertia"][0][0]
        kind, prm, _, t = om["shapes"][0]
        if kind == "sphere":
            d.npc_n_spheres = 1
            d.npc_sphere_radius[0] = prm
            for k in range(3):
                d.npc_sphere_center[0][k] = t[k]
        else:  # upright cylinder (radius, length) -> two stacked spheres spanning the same height
            rad, length = prm
            d.npc_n_spheres = 2
            for i, sgn in enumerate((-1.0, 1.0)):
                d.npc_sphere_radius[i] = rad
                d.npc_sphere_center[i][0], d.npc_sphere_center[i][1] = t[0], t[1]
                d.npc_sphere_center[i][2] = t[2] + sgn * max(0.0, length / 2 - rad)
    d.npc_contact_cap = 2
    if d.npc_kind == abi.NPC["box"]:
        # free box (box.urdf: 1 x 1 x 1 m, 6 kg): the robots' spheres collide with the oriented box itself; against the
        # terrain it is represented by its 8 corners (spheres of radius 2 cm inset by their radius)
        om = urdf_model.load_model("box", resources_root)["bodies"][0]
        d.npc_mass, d.npc_inertia = om["mass"], om["inertia"][0][0]
        kind, half, _, t = om["shapes"][0]
        assert kind == "box"
        rc = 0.02
        d.npc_n_spheres = 8
        for i in range(8):
            d.npc_sphere_radius[i] = rc
            for k in range(3):
                d.npc_sphere_center[i][k] = t[k] + (1.0 if (i >> k) & 1 else -1.0) * (half[k] - rc)
        for k in range(3):
            d.npc_box_half[k] = half[k]
        d.npc_contact_cap = 4
    d.seesaw_axis = 1
    d.n_static_boxes, d.npc_reported_bodies = 0, 1
    if d.npc_kind == abi.NPC["bridge"]:
        # fixed-base scenery (bridge.urdf: deck + two end blocks; wrestling.urdf: the raised field): the collision meshes
        # are SolidWorks boxes and, after the 90-degree joint rotations, world-aligned; the 3 cm painted rings of the
        # wrestling field (half height < 2 cm) are left out
        name = cfg.asset.name_npc
        scenery = urdf_model.load_model(name, resources_root)["bodies"][0]
        d.npc_reported_bodies = len(scenery["shapes"])           # one rigid body per URDF link (fixed joints are not merged)
        n = 0
        for kind, half, R, t in scenery["shapes"]:
            Rm = np.abs(np.asarray(R, np.float64))
            assert kind == "box" and np.allclose(Rm, np.round(Rm), atol=1e-3), "scenery boxes must be world-aligned"
            hw = Rm @ np.asarray(half, np.float64)
            if hw[2] < 0.02:
                continue
            for k in range(3):
                d.static_box_center[n][k], d.static_box_half[n][k] = t[k], hw[k]
            n += 1
        assert 1 <= n <= 4
        d.n_static_boxes = n
    if getattr(cfg.asset, "name_npc", "") == "rotation":
        # revolving door (rotation_door.urdf): fixed base disk + one box link on a vertical hinge, no joint range, no drive.
        # It runs on the seesaw code path (same articulation) with the hinge axis switched to +z; the 4 cm base disk is
        # represented by its inscribed square as the static "platform" box, there is no column.
        bodies = urdf_model.load_model("rotation", resources_root)["bodies"]
        base, door = bodies[0], bodies[1]
        for k in range(3):
            d.seesaw_joint_offset[k] = door["joint_offset"][k]
            d.seesaw_plank_center[k] = door["shapes"][0][3][k]
            d.seesaw_plank_half[k] = door["shapes"][0][1][k]
        rad, length = base["shapes"][0][1]
        d.seesaw_base_half[0] = d.seesaw_base_half[1] = rad / math.sqrt(2.0)
        d.seesaw_base_half[2] = length / 2
        d.seesaw_plank_mass, d.seesaw_plank_inertia_yy = door["mass"], door["inertia"][2][2]
        d.seesaw_vel_limit = door["velocity"]
        d.seesaw_default_angle = 0.0
        d.seesaw_column_radius = d.seesaw_column_length = 0.0
        d.seesaw_theta_lo, d.seesaw_theta_hi = -1e9, 1e9
        d.seesaw_axis = 2
    elif getattr(cfg.asset, "name_npc", "") == "circular":
        # tug-of-war disc (cylinder.urdf): fixed 1 mm base + an upright cylinder (r 1.2, collision height 0.5) on a prismatic
        # +y joint, range +-10 m, velocity limit 1 m/s, no drive: the 1-dof link path again, with a translating link
        bodies = urdf_model.load_model("circular", resources_root)["bodies"]
        disc = bodies[1]
        kind, (rad, length), _, tc = disc["shapes"][0]
        assert kind == "cylinder"
        for k in range(3):
            d.seesaw_joint_offset[k] = disc["joint_offset"][k]
            d.seesaw_plank_center[k] = tc[k]
            d.seesaw_base_half[k] = 0.0
        d.seesaw_plank_half[0] = d.seesaw_plank_half[1] = rad
        d.seesaw_plank_half[2] = length / 2
        d.seesaw_plank_mass = d.seesaw_plank_inertia_yy = disc["mass"]      # generalized inertia of a slider = its mass
        d.seesaw_vel_limit = disc["velocity"]
        d.seesaw_default_angle = 0.0
        d.seesaw_column_radius = d.seesaw_column_length = 0.0
        d.seesaw_theta_lo, d.seesaw_theta_hi = disc["lower"], disc["upper"]
        d.seesaw_axis, d.seesaw_link_cylinder = 3, 1
    elif d.npc_kind == abi.NPC["seesaw"]:
        bodies = urdf_model.load_model("seesaw", resources_root)["bodies"]
        base, plank = bodies[0], bodies[1]
        for k in range(3):
            d.seesaw_joint_offset[k] = plank["joint_offset"][k]
            d.seesaw_plank_center[k] = plank["shapes"][0][3][k]
            d.seesaw_plank_half[k] = plank["shapes"][0][1][k]
            d.seesaw_base_half[k] = base["shapes"][0][1][k]
        d.seesaw_plank_mass, d.seesaw_plank_inertia_yy = plank["mass"], plank["inertia"][1][1]
        d.seesaw_vel_limit = plank["velocity"]
        d.seesaw_default_angle = getattr(cfg.init_state, "default_npc_joint_angles", [0.0])[0]
        col = base["shapes"][1]                       # ("cylinder", (radius, length), R, t)
        d.seesaw_column_radius, d.seesaw_column_length = col[1][0], col[1][1]
        # The plank's COM sits on the hinge (seesaw.urdf:41-47) so gravity exerts no torque; what bounds its swing is an
        # end touching the ground slab.  The URDF gives no joint range (lower = upper = 0 by omission, while the task
        # starts it at -0.2 rad), so the engine uses the geometric stops.
        pivot_z = cfg.init_state.init_states_npc[0].pos[2] + plank["joint_offset"][2]
        cx, hx, hz = plank["shapes"][0][3][0], plank["shapes"][0][1][0], plank["shapes"][0][1][2]
        clear = pivot_z - hz - terrain.ground_z
        d.seesaw_theta_lo = -math.asin(min(1.0, clear / (hx - cx)))    # -x end (longer arm) down
        d.seesaw_theta_hi = math.asin(min(1.0, clear / (hx + cx)))     # +x end down
    # control
    ctl = cfg.control
    d.control_type = abi.CTRL[ctl.control_type]
    d.action_scale, d.hip_scale_reduction = ctl.action_scale, getattr(ctl, "hip_scale_reduction", 1.0)
    d.clip_actions = cfg.normalization.clip_actions
    names = m["dof_names"]
    tl = getattr(ctl, "torque_limits", None)
    for j in range(abi.NDOF):
        d.torque_limits[j] = (tl[j % len(tl)] if isinstance(tl, (list, tuple)) else tl) if tl is not None else m["dof_effort"][j]
        d.default_dof_pos[j] = cfg.init_state.default_joint_angles[names[j]]
    d.kp = next((v for k, v in ctl.stiffness.items() if k in names[0]), 0.0)
    d.kd = next((v for k, v in ctl.damping.items() if k in names[0]), 0.0)
    cmd, idx = fill_command_obs(cfg)
    if "gait" in idx:
        # the reference raises in preprocess_action itself (go1.py:76-77)
        raise NotImplementedError("command.cfg.gait: not implemented upstream either (go1.py:76-77 raises NotImplementedError)")
    # Go1.preprocess_action (go1.py:64-93): which action column feeds which entry of the locomotion observation.  Shipped configs:
    # only the velocity command (entries 3-5 <- columns 0-2); further flags turn entries 6-17 from constants of the scene into inputs.
    sc = ctl.obs_scales
    slots = {"vel": ((3, sc.lin_vel), (4, sc.lin_vel), (5, sc.ang_vel)), "body_height": ((6, sc.body_height),), "gait_freq": ((7, sc.gait_freq),),
             "footswing_height": ((12, sc.footswing_height),), "body_pose": ((13, sc.body_pitch), (14, sc.body_roll)),
             "stance_width": ((15, sc.stance_width),), "stance_length": ((16, sc.stance_length),), "aux_reward": ((17, sc.aux_reward),)}
    for c in range(18):
        d.command_src[c], d.command_scale[c] = -1, 0.0
    width = 0
    for flag, first in idx.items():
        for j, (c, scale) in enumerate(slots[flag]):
            d.command_src[c], d.command_scale[c] = first + j, float(scale)
        width = max(width, first + len(slots[flag]))
    if width == 0:
        raise NotImplementedError("command.cfg switches every command off: Go1.step would take actions of width 0")
    d.num_command_dims = width
    for k in range(70):
        d.command_obs[k] = cmd[k]
    d.cmd_lin_scale, d.cmd_ang_scale = ctl.obs_scales.lin_vel, ctl.obs_scales.ang_vel
    d.clip_command = 0 if task == "football_defender" else 1
    # terrain
    d.wall_sdf = _fp(terrain.wall_sdf, keep)
    d.sdf_nx, d.sdf_ny = terrain.wall_sdf.shape
    d.horizontal_scale, d.wall_height, d.ground_z = cfg.terrain.horizontal_scale, terrain.wall_height, terrain.ground_z
    gh = getattr(terrain, "ground_height", None)              # Perlin relief of the walkable surface (None: flat slab)
    if gh is not None:
        assert gh.shape == terrain.wall_sdf.shape
        d.ground_height = _fp(gh, keep)
    wt = getattr(terrain, "wall_top", None)                   # per-block wall heights (None: one height, d.wall_height)
    if wt is not None:
        assert wt.shape == terrain.wall_sdf.shape
        d.wall_top = _fp(wt, keep)
    # edge contacts (include/mqe_hip.h edge_contacts): on by default; MQE_EDGE_CONTACTS=<mask> overrides (0: round 3's feature-point tests only)
    d.edge_contacts = int(os.environ.get("MQE_EDGE_CONTACTS", edge_contacts if edge_contacts is not None else 3))        # (bit 4, box edges against box primitives: available, off by default)
    if os.environ.get("MQE_CONTACT_REDUCTION", "0") not in ("0", ""):      # bit 8: manifold reduction of a robot's one-sided contacts to its deepest eight (off by default)
        d.edge_contacts |= 8
    wc = getattr(terrain, "wall_corner", None)
    if wc is not None and (d.edge_contacts & 1):
        assert wc.shape == terrain.wall_sdf.shape + (2,)
        d.wall_corner = _fp(wc, keep)
    d.soft_dof_pos_limit = float(getattr(cfg.rewards, "soft_dof_pos_limit", 1.0))
    d.env_origins = _fp(env_origins, keep)
    # run-time terrain curriculum (legged_robot.py:479-503): the per-track origin table and each env's (level, type) at construction
    if getattr(cfg.terrain, "curriculum", False) and int(cfg.terrain.num_rows) > 1 and terrain_levels is not None:
        tab = np.ascontiguousarray(terrain.env_origins, np.float32)
        assert tab.shape == (cfg.terrain.num_rows, cfg.terrain.num_cols, 3), tab.shape
        lv, ty = np.ascontiguousarray(terrain_levels, np.int32), np.ascontiguousarray(terrain_types, np.int32)
        assert lv.shape == (num_envs,) and ty.shape == (num_envs,)
        keep += [tab, lv, ty]
        d.terrain_curriculum, d.terrain_num_rows, d.terrain_num_cols = 1, int(cfg.terrain.num_rows), int(cfg.terrain.num_cols)
        d.terrain_env_length = float(terrain.env_length)
        d.terrain_origins = tab.ctypes.data_as(abi.FP)
        d.terrain_levels = lv.ctypes.data_as(C.POINTER(C.c_int32))
        d.terrain_types = ty.ctypes.data_as(C.POINTER(C.c_int32))
    d.agent_origins = _fp(agent_origins, keep)
    st = cfg.init_state
    if getattr(st, "multi_init_state", False):
        base = [s.pos + s.rot + s.lin_vel + s.ang_vel for s in st.init_states]
    else:
        base = [st.pos + st.rot + st.lin_vel + st.ang_vel] * A
    assert len(base) == A, "need one init state per agent"
    d.base_init_state = _fp(np.asarray(base, np.float32), keep)
    if P:
        npc_init = _npc_init_states(cfg, P)
        d.npc_init_state = _fp(npc_init, keep)
    if gate_pos is not None:
        d.gate_pos = _fp(gate_pos, keep)
    # termination
    dt_policy = ctl.decimation * cfg.sim.dt
    d.max_episode_length = int(np.ceil(cfg.env.episode_length_s / dt_policy))
    tm = getattr(cfg, "termination", None)
    flags = 0
    if tm is not None:
        for t in tm.termination_terms:
            flags |= abi.TERM[t]
        d.roll_threshold, d.pitch_threshold = tm.roll_kwargs["threshold"], tm.pitch_kwargs["threshold"]
        d.z_low_threshold, d.z_high_threshold = tm.z_low_kwargs["threshold"], tm.z_high_kwargs["threshold"]
    d.termination_flags = flags
    d.terminate_on_base_contact = 1 if len(cfg.asset.terminate_after_contacts_on) else 0
    # reset distribution
    dr = cfg.domain_rand
    d.noise_mode = noise_mode
    rr = getattr(dr, "init_dof_pos_ratio_range", None) or [1.0, 1.0]
    d.dof_ratio_lo, d.dof_ratio_hi = rr
    bp = getattr(dr, "init_base_pos_range", None)
    if bp is not None:
        d.has_base_pos_range = 1
        d.base_pos_x_lo, d.base_pos_x_hi = bp["x"]
        d.base_pos_y_lo, d.base_pos_y_hi = bp["y"]
    npr = getattr(dr, "init_npc_base_pos_range", None)
    if npr is not None and P:
        d.has_npc_pos_range = 1
        d.npc_pos_x_lo, d.npc_pos_x_hi = npr["x"]
        d.npc_pos_y_lo, d.npc_pos_y_hi = npr["y"]
    bv = getattr(dr, "init_base_vel_range", None) or (-0.5, 0.5)
    d.base_vel_lo, d.base_vel_hi = bv
    # domain randomisation switches (legged_robot.py:283-336, legged_robot_field.py:324-334, go1.py:237,337)
    if getattr(dr, "randomize_friction", False):
        d.rand_friction = 1
        d.friction_lo, d.friction_hi = dr.friction_range
    if getattr(dr, "randomize_base_mass", False):
        d.rand_base_mass = 1
        d.added_mass_lo, d.added_mass_hi = dr.added_mass_range
    if getattr(dr, "randomize_com", False):
        d.rand_com = 1
        for k, ax in enumerate("xyz"):
            d.com_lo[k], d.com_hi[k] = getattr(dr.com_range, ax)
    if getattr(dr, "randomize_lag_timesteps", False):
        d.lag_timesteps = int(dr.lag_timesteps)
    if getattr(dr, "push_robots", False):
        d.push_interval = int(np.ceil(dr.push_interval_s / (cfg.sim.dt * cfg.control.decimation)))     # legged_robot.py:1024
        d.max_push_vel_xy = float(dr.max_push_vel_xy)
    d.sheep_movement_scale = getattr(cfg.asset, "sheep_movement_scale", 0.0)
    d.sheep_movement_randomness = getattr(cfg.asset, "sheep_movement_randomness", 0.0)
    for i, (attr, _) in enumerate(REWARD_TERMS[task]):
        if attr is not None:
            d.reward_scale[i] = float(getattr(cfg.rewards.scales, attr, 0.0))
    kw = cfg.terrain.BarrierTrack_kwargs
    if task == "pushbox":
        d.reward_scale[0] = 1.0       # hard-set in the wrapper's constructor after the cfg value was copied (:20)
    if task == "rotation":            # likewise (go1_rotation_wrapper.py:18-20); the target x = 0.75 * rotation block + wall (:32-35)
        d.reward_scale[0], d.reward_scale[1], d.reward_scale[2] = 5.0, 1.0, 1.0
        d.wrapper_param[0] = kw["rotation"]["block_length"] * 0.75 + kw["wall"]["block_length"]
    if task == "gate":
        d.wrapper_param[0] = kw["init"]["block_length"] + kw["gate"]["block_length"] + kw["plane"]["block_length"] / 2
        d.wrapper_param[1] = kw["track_width"] / 4
    # networks
    aW, ab = policy_weights.load_actuator_net()
    dW, db = policy_weights.load_adaptation_module()
    if body is None:
        bW, bb, synthetic = policy_weights.load_body(getattr(ctl, "locomotion_policy_dir", None), seed=0)
    else:
        bW, bb = body
        synthetic = False
    _fill_mlp(d.actuator, aW, ab, keep)
    _fill_mlp(d.adaptation, dW, db, keep)
    _fill_mlp(d.body, bW, bb, keep)
    keep.append(("body_is_synthetic", synthetic))
    return d, keep


def _npc_init_states(cfg, P):
    """Per-NPC initial root state.  ball/seesaw: cfg.init_state.init_states_npc (go1_object.py:27-51);
    sheep: a num_rows x num_cols grid centred on the second block (go1_sheep.py:84-118) -- the reference also
    draws a random yaw per sheep from np.random there; we draw the same way so seeded runs agree."""
    if hasattr(cfg.init_state, "init_states_npc"):
        return np.asarray([s.pos + s.rot + s.lin_vel + s.ang_vel for s in cfg.init_state.init_states_npc], np.float32)
    a, kw = cfg.asset, cfg.terrain.BarrierTrack_kwargs
    nr, nc, dis = a.num_rows, a.num_cols, a.dis_sheep
    origin = np.array([kw["init"]["block_length"] + kw["plane"]["block_length"] / 2 - nr // 2 * dis[0], -(nc // 2) * dis[1], 0.3])
    pos = origin.copy()
    out = []
    for i in range(nr):
        for j in range(nc):
            rot = np.array([0.0, 0.0, 0.0, 1.0]) + np.random.randn(4) * np.array([0, 0, math.pi, 1])
            rot = rot / np.linalg.norm(rot)   # the reference hands the raw 4-vector to PhysX, which normalises it
            out.append(np.concatenate((pos, rot, np.zeros(3), np.zeros(3))))
            pos[1] += dis[1]
        pos[0] += dis[0]
        pos[1] = origin[1]
    assert len(out) == P
    return np.asarray(out, np.float32)
