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
 * Perlin relief (`add_perlin_noise` + `border_perlin_noise`, reference :372-393 with perlin.py:33-72) and curriculum rows
   (`cfg.curriculum`, :421-439,635-638) are generated with the reference's random stream; the engine receives the relief of the
   walkable surface as a second map, `ground_height` [m] at the SDF's raster points (entry (i, j) at the world point (i hs, j hs), the vertices of upstream's trimesh), next to the wall set.  As upstream, the
   per-track noise of `add_track_to_sim` is drawn (it advances `np.random`) but only the whole-map noise of
   `build_heightfield_raw` ever reaches the heightfield: a track keeps it where its noise mask is 1 (:449-459).
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


def perlin_octave(samples, cells):
    """One octave of gradient noise on a (samples[0], samples[1]) raster with `cells` lattice cells per axis, values ~[0, 1]
    (reference perlin.py:33-57).  One unit gradient per lattice node from `np.random.rand`; a raster point k of an axis lies in
    cell k // (samples // cells) at the fractional position (k * cells / samples) mod 1; quintic fade 6t^5 - 15t^4 + 10t^3."""
    nx, ny = int(samples[0]), int(samples[1])
    cx, cy = int(cells[0]), int(cells[1])
    assert nx % cx == 0 and ny % cy == 0, "raster must be a whole number of lattice cells (as upstream's repeat() requires)"
    theta = 2 * np.pi * np.random.rand(cx + 1, cy + 1)
    gx, gy = np.cos(theta), np.sin(theta)
    ix = np.arange(nx) // (nx // cx)
    iy = np.arange(ny) // (ny // cy)
    # np.mgrid[0:c:c/n] is start + k * step in double precision; its length must come out as n for upstream to run at all
    fx = (np.arange(nx) * (cx / nx)) % 1
    fy = (np.arange(ny) * (cy / ny)) % 1
    FX, FY = fx[:, None], fy[None, :]
    IX, IY = ix[:, None], iy[None, :]

    def corner(dx, dy):          # gradient of the cell corner (dx, dy) dotted with the offset from that corner
        return (FX - dx) * gx[IX + dx, IY + dy] + (FY - dy) * gy[IX + dx, IY + dy]
    fade_x = 6 * FX ** 5 - 15 * FX ** 4 + 10 * FX ** 3
    fade_y = 6 * FY ** 5 - 15 * FY ** 4 + 10 * FY ** 3
    low = corner(0, 0) * (1 - fade_x) + fade_x * corner(1, 0)
    high = corner(0, 1) * (1 - fade_x) + fade_x * corner(1, 1)
    return np.sqrt(2) * ((1 - fade_y) * low + fade_y * high) * 0.5 + 0.5


def fractal_noise(xSize=20, ySize=20, xSamples=1600, ySamples=1600, frequency=10, fractalOctaves=2, fractalLacunarity=2.0,
                  fractalGain=0.25, zScale=0.23):
    """Sum of octaves (reference perlin.py:59-72): lattice `frequency` cells per metre, each octave `fractalLacunarity` x finer
    and `fractalGain` x weaker; metres."""
    cx, cy = int(frequency * xSize), int(frequency * ySize)
    out = np.zeros((xSamples, ySamples))
    amp = 1
    for _ in range(fractalOctaves):
        out += amp * perlin_octave((xSamples, ySamples), (cx, cy)) * zScale
        amp *= fractalGain
        cx, cy = int(fractalLacunarity * cx), int(fractalLacunarity * cy)
    return out


class BarrierTrack:
    def __init__(self, cfg, num_envs: int, num_agents=1):
        self.cfg = cfg
        self.num_envs = num_envs
        self.num_agents = num_agents
        assert cfg.mesh_type == "trimesh", "BarrierTrack needs cfg.terrain.mesh_type == 'trimesh'"
        assert getattr(cfg, "BarrierTrack_kwargs", None) is not None, "cfg.terrain.BarrierTrack_kwargs missing"
        self.track_kwargs = dict(_DEFAULTS)
        self.track_kwargs.update(cfg.BarrierTrack_kwargs)
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
        return np.full(res, H, dtype=np.float32), None, {}, np.zeros(res, np.float32)

    def get_plane_block(self, thickness, res):
        H = _pick(self.track_kwargs["wall_height"]) / self.cfg.vertical_scale
        h = np.zeros(res, dtype=np.float32)
        tpx = self._px(thickness)
        self._side_walls(h, H, tpx)
        mask = np.zeros(res, np.float32)
        mask[:, tpx: res[1] - tpx] = 1.0
        return h, None, {}, mask

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
        mask = np.zeros(res, np.float32)
        h[: x0 + room[0], :] = H                      # everything behind the start rooms is solid
        self._side_walls(h, H, tpx)
        mask[x0 + room[0]:, tpx: res[1] - tpx] = 1.0
        spawn = np.zeros((n, 3), dtype=np.float32)
        for i in range(n):
            ya = y0 + i * (room[1] + gap)
            h[x0: x0 + room[0], ya: y0 + (i + 1) * room[1] + i * gap] = 0.0
            mask[x0: x0 + room[0], ya: y0 + (i + 1) * room[1] + i * gap] = 1.0
            spawn[i, 0] = x0 + int(room[0] / 2)
            spawn[i, 1] = ya + int(room[1] / 2)
        self._side_walls(h, H, tpx)
        h[:tpx, :] = H
        return h, spawn, {}, mask

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
        mask = np.ones(res, np.float32)
        h[org[0]: org[0] + dpx, :] = H
        self._side_walls(h, H, tpx)
        mask[org[0]: org[0] + dpx, :] = 0.0
        mask[:, :tpx] = 0.0
        mask[:, -tpx:] = 0.0
        h[org[0]: org[0] + dpx, org[1]: org[1] + wpx] = 0.0
        mask[org[0]: org[0] + dpx, org[1]: org[1] + wpx] = 1.0
        return h, None, {"gate_deviation": (off + jitter).astype(np.float32) * hs}, mask

    # -- assembly --------------------------------------------------------------------------------------
    def _perlin_kwargs(self, difficulty, first_only):
        """TerrainPerlin_kwargs with its (lo, hi) entries resolved: the whole-map noise takes lo (reference :376-381); a track draws
        U(lo, hi), or interpolates by its row's difficulty under `curriculum_perlin`, and drops values below `no_perlin_threshold`
        (:423-434)"""
        kw = dict(getattr(self.cfg, "TerrainPerlin_kwargs", {}) or {})
        for k, v in list(kw.items()):
            if isinstance(v, (tuple, list)):
                if first_only:
                    kw[k] = v[0]
                else:
                    if difficulty is None or not self.track_kwargs["curriculum_perlin"]:
                        kw[k] = np.random.uniform(*v)
                    else:
                        kw[k] = v[0] * (1 - difficulty) + v[1] * difficulty
                    if self.track_kwargs["no_perlin_threshold"] > kw[k]:
                        kw[k] = 0.0
        return kw

    def build(self):
        cfg, hs, kwt = self.cfg, self.cfg.horizontal_scale, self.track_kwargs
        self.initialize_track()
        self.border = int(cfg.border_size / hs)
        self.tot_rows = int(cfg.num_rows * self.track_resolution[0]) + 2 * self.border
        self.tot_cols = int(cfg.num_cols * self.track_resolution[1]) + 2 * self.border
        hf = np.zeros((self.tot_rows, self.tot_cols), dtype=np.float32)
        wall = np.zeros(hf.shape, dtype=bool)          # pixels raised by a block painter / the border: the engine's wall set
        relief = None                                  # whole-map noise [vertical units], kept unmasked for the engine's ground map
        perlin_map = bool(kwt["add_perlin_noise"] and kwt["border_perlin_noise"])
        heights = set()                                # distinct wall heights [vertical units]: block walls and the raised border
        if perlin_map:                                 # build_heightfield_raw (:372-393)
            relief = fractal_noise(xSize=self.env_length * cfg.num_rows + 2 * cfg.border_size,
                                   ySize=self.env_width * cfg.num_cols + 2 * cfg.border_size,
                                   xSamples=self.tot_rows, ySamples=self.tot_cols, **self._perlin_kwargs(None, True)) / cfg.vertical_scale
            hf += relief
            if kwt["border_height"] != 0.0 and self.border > 0:
                hf[:, :self.border] += kwt["border_height"] / cfg.vertical_scale
                hf[:, -self.border:] += kwt["border_height"] / cfg.vertical_scale
                if kwt["border_height"] > 0:
                    wall[:, :self.border] = True
                    wall[:, -self.border:] = True
                    heights.add(kwt["border_height"] / cfg.vertical_scale)     # the border is a wall of its own height (relief + border_height)
        self.track_origins_px = np.zeros((cfg.num_rows, cfg.num_cols, 3), dtype=int)
        self.track_width_map = np.zeros((cfg.num_rows, cfg.num_cols), dtype=np.float32)
        infos = {}
        wall_px = np.zeros_like(hf)                      # height [px units] of the wall standing on each pixel (0: none)
        for c in range(cfg.num_cols):
            for r in range(cfg.num_rows):
                org = np.array([int(r * self.track_resolution[0]) + self.border, int(c * self.track_resolution[1]) + self.border, 0])
                self.track_origins_px[r, c] = org
                difficulty = r / (cfg.num_rows - 1) if (getattr(cfg, "curriculum", False) and cfg.num_rows > 1) else None   # get_difficulty (:635-638)
                if kwt["add_perlin_noise"]:
                    # add_track_to_sim draws a noise field per track (:421-439) and never adds it (:449-459 only re-uses what is
                    # already in heightfield_raw); the draw is repeated here because it advances np.random
                    fractal_noise(xSize=self.env_length, ySize=self.env_width, xSamples=self.track_resolution[0],
                                  ySamples=self.track_resolution[1], **self._perlin_kwargs(difficulty, False))
                thickness = _pick(kwt["wall_thickness"])
                x = org[0]
                spawn = None
                for bi, name in enumerate(kwt["options"]):
                    res = self.track_block_resolutions[bi]
                    h, sp, info, mask = getattr(self, "get_" + name + "_block")(thickness, res)
                    sl = (slice(x, x + res[0]), slice(org[1], org[1] + res[1]))
                    hf[sl] = h + mask * hf[sl] + org[2]
                    wall[sl] = h > 0
                    wall_px[sl] = h
                    heights.update(np.unique(h[h > 0]).tolist())
                    x += res[0]
                    if sp is not None:
                        assert spawn is None, "a track may contain one init block only"
                        spawn = sp
                    for k, v in info.items():
                        infos.setdefault(k, np.zeros((cfg.num_rows, cfg.num_cols, v.shape[-1]), np.float32))[r, c] = v
                self.track_width_map[r, c] = self.env_width - thickness * 2
                self.agent_origins[r, c, :, :2] = (org[None, :2] + spawn[:, :2]) * hs
                self.agent_origins[r, c, :, 2] = (org[2] + spawn[:, 2]) * cfg.vertical_scale
                self.env_origins[r, c] = [org[0] * hs, org[1] * hs + kwt["track_width"] / 2, org[2] * cfg.vertical_scale]
        self.heightfield_raw = hf
        self.heightsamples = hf
        self.env_info = infos
        # one wall height per scene is a scalar; a (lo, hi) wall_height draws one per block (:167-173,191-199,218-239) and the engine
        # then gets a map: at every raster point the top of the wall nearest to it
        self.wall_height = float(max(heights) * cfg.vertical_scale) if heights else 0.0
        self.wall_top = None
        if len(heights) > 1:
            wall_px[wall & (wall_px <= 0)] = max(kwt["border_height"], 0.0) / cfg.vertical_scale      # raised Perlin borders
            _, (ii, jj) = ndimage.distance_transform_edt(~(wall_px > 0), return_indices=True)
            self.wall_top = (wall_px[ii, jj] * cfg.vertical_scale).astype(np.float32)
        # flat scenes lie under the reference's 2 cm ground slab (:628-632); with the Perlin map there is no slab (:567-626): the
        # ground is the heightfield itself
        self.ground_z = 0.0 if perlin_map else GROUND_SLAB_TOP
        self.ground_height = (relief * cfg.vertical_scale).astype(np.float32) if perlin_map else None
        self.wall = wall
        self.wall_sdf = self._signed_distance(wall, hs)
        self.wall_corner = self._nearest_convex_corner(wall, hs)
        return self

    @staticmethod
    def _nearest_convex_corner(wall, hs):
        """(nx, ny, 2) float32: for every raster point the world (x, y) of the nearest CONVEX corner of the wall pixel set -- the vertical
        edges of the wall prisms (a pixel = the hs x hs square centred on its raster point, so corners sit on the half-integer lattice).
        What the engine tests the robots' primitives against between their feature points: a gate post's corner pressing into the side
        of the trunk.  None when the set has no convex corner."""
        W = np.pad(np.asarray(wall, bool), 1)
        q = W[:-1, :-1].astype(np.int8) + W[1:, :-1] + W[:-1, 1:] + W[1:, 1:]      # wall pixels around lattice node (a, b) = world ((a - .5) hs, (b - .5) hs)
        convex = q == 1
        if not convex.any():
            return None
        _, (ia, ib) = ndimage.distance_transform_edt(~convex, return_indices=True)
        nx, ny = wall.shape
        ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
        best, bd = None, None
        for da in (0, 1):                      # the four lattice nodes around a raster point: the nearest of their nearest corners
            for db in (0, 1):
                ca, cb = ia[ii + da, jj + db], ib[ii + da, jj + db]
                dist = (ca - 0.5 - ii) ** 2 + (cb - 0.5 - jj) ** 2
                cand = np.stack([(ca - 0.5) * hs, (cb - 0.5) * hs], -1)
                if best is None:
                    best, bd = cand, dist
                else:
                    m = dist < bd
                    best[m], bd[m] = cand[m], dist[m]
        return np.ascontiguousarray(best, np.float32)

    def add_terrain_to_sim(self, gym=None, sim=None, device="cpu"):
        """Name kept for source compatibility (reference barrier_track.py:501); gym/sim are ignored."""
        self.device = device
        return self.build()

    @staticmethod
    def _signed_distance(wall, hs):
        """[m] distance from each raster point to the wall pixel set (negative inside), exact EDT; a pixel is the hs x hs square
        centred on its raster point (i hs, j hs): see map_sample in oracle/mqe_oracle.c for what that approximates upstream."""
        if not wall.any():
            return np.full(wall.shape, 1e3, np.float32)
        outside = ndimage.distance_transform_edt(~wall) - 0.5
        inside = ndimage.distance_transform_edt(wall) - 0.5
        return (np.where(wall, -inside, outside) * hs).astype(np.float32)
