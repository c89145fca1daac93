"""BarrierTrack: rows x cols grid of straight tracks assembled from blocks ("init", "gate", "plane", "wall"),
rasterised to a heightfield at `horizontal_scale` metres per pixel.

Restates the generator the reference runs at scene-construction time (mqe/utils/terrain/barrier_track.py:
initialize_track :90-122, the four block painters :157-262 and :311-362, grid assembly :501-565, flat ground slab
:567-632).  Outputs match the reference for the same `np.random` seed (tests/test_terrain.py against
tests/golden/terrain_*.npz): `heightfield_raw`, `env_origins`, `agent_origins`, `env_info["gate_deviation"]`.

Differences by design:
 * instead of a PhysX triangle mesh the engine collides against (a) the ground slab top at z = 0.02 m (the
   reference's plane box, :628-632, lies over the whole map) and (b) the wall set W = {pixels with height > 0},
   handed over as a 2-D signed distance field `wall_sdf` sampled at pixel centres (DESIGN.md "terrain");
 * `track_kwargs` is per instance (the reference mutates a class-level dict, :13-53,62, leaking state between
   environments created in one process);
 * "rotation" blocks, Perlin noise and curriculum difficulty are not implemented (unused by the BASELINE configs).
"""
import numpy as np
from scipy import ndimage

_DEFAULTS = dict(
    options=["gate", "init", "wall", "plane"], track_width=1.6, track_length=None, wall_thickness=0.04,
    wall_height=0.5, wall=dict(block_length=3.0), plane=dict(block_length=3.0),
    init=dict(block_length=1.2, room_size=(0.8, 0.8), border_with=0.05, offset=(0, 0)),
    gate=dict(block_length=1.2, width=1.0, depth=1.0, offset=(0, 0)),
    add_perlin_noise=False, border_perlin_noise=False, border_height=0.0, virtual_terrain=False,
    check_skill_combinations=False, engaging_next_threshold=0.0, curriculum_perlin=True, no_perlin_threshold=0.02,
)

GROUND_SLAB_TOP = 0.02  # [m] top face of the ground box (reference barrier_track.py:630-632)


def _pick(v):
    return np.random.uniform(*v) if isinstance(v, (tuple, list)) else v


class BarrierTrack:
    def __init__(self, cfg, num_envs: int, num_agents=1):
        self.cfg = cfg
        self.num_envs = num_envs
        self.num_agents = num_agents
        assert cfg.mesh_type == "trimesh", "BarrierTrack needs cfg.terrain.mesh_type == 'trimesh'"
        assert getattr(cfg, "BarrierTrack_kwargs", None) is not None, "cfg.terrain.BarrierTrack_kwargs missing"
        self.track_kwargs = dict(_DEFAULTS)
        self.track_kwargs.update(cfg.BarrierTrack_kwargs)
        if self.track_kwargs["add_perlin_noise"]:
            raise NotImplementedError("Perlin-noise tracks are out of scope (SURVEY.md 8f rank 4)")
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3), dtype=np.float32)
        self.agent_origins = np.zeros((cfg.num_rows, cfg.num_cols, num_agents, 3), dtype=np.float32)
        self.env_info = None

    # -- sizes ---------------------------------------------------------------------------------------
    def _px(self, metres):
        return np.ceil(metres / self.cfg.horizontal_scale).astype(int)

    def initialize_track(self):
        kw, hs = self.track_kwargs, self.cfg.horizontal_scale
        self.env_block_lengths = [kw[o]["block_length"] for o in kw["options"]]
        length = 0.0
        for bl in self.env_block_lengths:
            length += bl
        kw["track_length"] = length
        wpx = np.ceil(kw["track_width"] / hs).astype(int)
        self.track_block_resolutions = [(np.ceil(bl / hs).astype(int), wpx) for bl in self.env_block_lengths]
        self.track_resolution = (np.ceil(length / hs).astype(int), wpx)
        self.n_blocks_per_track = len(kw["options"])
        self.env_length, self.env_width = length, kw["track_width"]

    # -- block painters: return (heights [px], keep_mask, agent_spawn_px or None, info) ----------------
    def _side_walls(self, h, H, tpx):
        h[:, :tpx] = H
        h[:, -tpx:] = H

    def get_wall_block(self, thickness, res):
        H = _pick(self.track_kwargs["wall_height"]) / self.cfg.vertical_scale
        return np.full(res, H, dtype=np.float32), None, {}

    def get_plane_block(self, thickness, res):
        H = _pick(self.track_kwargs["wall_height"]) / self.cfg.vertical_scale
        h = np.zeros(res, dtype=np.float32)
        self._side_walls(h, H, self._px(thickness))
        return h, None, {}

    def get_init_block(self, thickness, res):
        kw, hs, n = self.track_kwargs["init"], self.cfg.horizontal_scale, self.num_agents
        H = _pick(self.track_kwargs["wall_height"]) / self.cfg.vertical_scale
        off = (int(kw["offset"][0] / hs), int(kw["offset"][1] / hs))
        room = (int(kw["room_size"][0] / hs), int(kw["room_size"][1] / hs))
        gap = np.ceil(kw["border_width"] / hs).astype(int)
        tpx = self._px(thickness)
        span_y = room[1] * n + gap * (n - 1)
        x0 = np.ceil((res[0] - room[0]) / 2).astype(int) + off[0]
        y0 = np.ceil((res[1] - span_y) / 2).astype(int) + off[1]
        h = np.zeros(res, dtype=np.float32)
        h[: x0 + room[0], :] = H                      # everything behind the start rooms is solid
        self._side_walls(h, H, tpx)
        spawn = np.zeros((n, 3), dtype=np.float32)
        for i in range(n):
            ya = y0 + i * (room[1] + gap)
            h[x0: x0 + room[0], ya: y0 + (i + 1) * room[1] + i * gap] = 0.0
            spawn[i, 0] = x0 + int(room[0] / 2)
            spawn[i, 1] = ya + int(room[1] / 2)
        self._side_walls(h, H, tpx)
        h[:tpx, :] = H
        return h, spawn, {}

    def get_gate_block(self, thickness, res):
        kw, hs = self.track_kwargs["gate"], self.cfg.horizontal_scale
        depth = _pick(kw["depth"])
        H = _pick(self.track_kwargs["wall_height"]) / self.cfg.vertical_scale
        off = np.asarray((np.ceil(kw["offset"][0] / hs).astype(int), np.ceil(kw["offset"][1] / hs).astype(int)))
        jitter = np.asarray((kw["random"][0] / hs, kw["random"][1] / hs)) if "random" in kw else np.zeros(2)
        jitter = np.ceil(jitter * (np.random.random(2) - 0.5) * 2).astype(int)
        width = _pick(kw["width"])
        dpx, wpx, tpx = int(depth / hs), int(width / hs), self._px(thickness)
        org = np.asarray([np.ceil((res[0] - dpx) / 2).astype(int), np.ceil((res[1] - wpx) / 2).astype(int)]) + off + jitter
        h = np.zeros(res, dtype=np.float32)
        h[org[0]: org[0] + dpx, :] = H
        self._side_walls(h, H, tpx)
        h[org[0]: org[0] + dpx, org[1]: org[1] + wpx] = 0.0
        return h, None, {"gate_deviation": (off + jitter).astype(np.float32) * hs}

    # -- assembly --------------------------------------------------------------------------------------
    def build(self):
        cfg, hs = self.cfg, self.cfg.horizontal_scale
        self.initialize_track()
        self.border = int(cfg.border_size / hs)
        self.tot_rows = int(cfg.num_rows * self.track_resolution[0]) + 2 * self.border
        self.tot_cols = int(cfg.num_cols * self.track_resolution[1]) + 2 * self.border
        hf = np.zeros((self.tot_rows, self.tot_cols), dtype=np.float32)
        self.track_origins_px = np.zeros((cfg.num_rows, cfg.num_cols, 3), dtype=int)
        self.track_width_map = np.zeros((cfg.num_rows, cfg.num_cols), dtype=np.float32)
        infos = {}
        for c in range(cfg.num_cols):
            for r in range(cfg.num_rows):
                org = np.array([int(r * self.track_resolution[0]) + self.border, int(c * self.track_resolution[1]) + self.border, 0])
                self.track_origins_px[r, c] = org
                thickness = _pick(self.track_kwargs["wall_thickness"])
                x = org[0]
                spawn = None
                for bi, name in enumerate(self.track_kwargs["options"]):
                    res = self.track_block_resolutions[bi]
                    h, sp, info = getattr(self, "get_" + name + "_block")(thickness, res)
                    hf[x: x + res[0], org[1]: org[1] + res[1]] = h
                    x += res[0]
                    if sp is not None:
                        assert spawn is None, "a track may contain one init block only"
                        spawn = sp
                    for k, v in info.items():
                        infos.setdefault(k, np.zeros((cfg.num_rows, cfg.num_cols, v.shape[-1]), np.float32))[r, c] = v
                self.track_width_map[r, c] = self.env_width - thickness * 2
                self.agent_origins[r, c, :, :2] = (org[None, :2] + spawn[:, :2]) * hs
                self.agent_origins[r, c, :, 2] = (org[2] + spawn[:, 2]) * cfg.vertical_scale
                self.env_origins[r, c] = [org[0] * hs, org[1] * hs + self.track_kwargs["track_width"] / 2, org[2] * cfg.vertical_scale]
        self.heightfield_raw = hf
        self.heightsamples = hf
        self.env_info = infos
        levels = np.unique(hf)
        if not (len(levels) <= 2 and levels[0] == 0.0):
            raise NotImplementedError("engine terrain model needs a two-level heightfield (floor + one wall height)")
        self.wall_height = float(levels[-1] * cfg.vertical_scale) if len(levels) == 2 else 0.0
        self.ground_z = GROUND_SLAB_TOP
        self.wall_sdf = self._signed_distance(hf > 0, hs)
        return self

    def add_terrain_to_sim(self, gym=None, sim=None, device="cpu"):
        """Name kept for source compatibility (reference barrier_track.py:501); gym/sim are ignored."""
        self.device = device
        return self.build()

    @staticmethod
    def _signed_distance(wall, hs):
        """[m] distance from each pixel centre to the wall pixel set (negative inside), exact EDT."""
        if not wall.any():
            return np.full(wall.shape, 1e3, np.float32)
        outside = ndimage.distance_transform_edt(~wall) - 0.5
        inside = ndimage.distance_transform_edt(wall) - 0.5
        return (np.where(wall, -inside, outside) * hs).astype(np.float32)
