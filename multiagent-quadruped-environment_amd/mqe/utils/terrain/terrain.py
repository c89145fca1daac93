"""Terrain: the legacy legged_gym height-field generator (reference mqe/utils/terrain/terrain.py:38-165; registry entry
mqe/utils/terrain/__init__.py:4), selectable through `cfg.terrain.selected = "Terrain"`.  No shipped task uses it.

The class body upstream is a dispatcher over `isaacgym.terrain_utils` -- a THIRD-PARTY module (NVIDIA Isaac Gym Preview 4,
python/isaacgym/terrain_utils.py) that is not part of the reference snapshot.  Its sub-terrain generators are restated below from
the published package; since nothing of it can be imported here, this restatement is UNPINNED (no reference vectors exist for
it; tests check shapes, determinism under np.random.seed and the geometric intent of each generator).  `gap_terrain` and
`pit_terrain` are the reference's own (terrain.py:167-192).

The int16 height samples reach the engine as the relief of the walkable surface (`ground_height`, bilinear between the raster
points = the vertices of upstream's mesh) next to an empty wall set; `slope_treshold` (upstream: steep faces become vertical in
the mesh) has no counterpart -- a stair riser is a one-cell ramp here."""
import numpy as np


class SubTerrain:
    """isaacgym.terrain_utils.SubTerrain: an int16 height raster of `width` x `length` samples"""

    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale, self.horizontal_scale = vertical_scale, horizontal_scale
        self.width, self.length = width, length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """heights drawn from a `step` ladder on a coarse grid, bilinearly upsampled, rounded, ADDED to the raster"""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    min_height, max_height, step = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale), int(step / terrain.vertical_scale)
    heights_range = np.arange(min_height, max_height + step, step)
    coarse = np.random.choice(heights_range, (int(terrain.width * terrain.horizontal_scale / downsampled_scale),
                                              int(terrain.length * terrain.horizontal_scale / downsampled_scale)))
    x = np.linspace(0, terrain.width * terrain.horizontal_scale, coarse.shape[0])
    y = np.linspace(0, terrain.length * terrain.horizontal_scale, coarse.shape[1])
    xu = np.linspace(0, terrain.width * terrain.horizontal_scale, terrain.width)
    yu = np.linspace(0, terrain.length * terrain.horizontal_scale, terrain.length)
    ix = np.clip(np.searchsorted(x, xu, side="right") - 1, 0, len(x) - 2)
    iy = np.clip(np.searchsorted(y, yu, side="right") - 1, 0, len(y) - 2)
    tx = ((xu - x[ix]) / (x[ix + 1] - x[ix]))[:, None]
    ty = ((yu - y[iy]) / (y[iy + 1] - y[iy]))[None, :]
    c = coarse.astype(np.float64)
    z = (c[ix][:, iy] * (1 - tx) * (1 - ty) + c[ix + 1][:, iy] * tx * (1 - ty) + c[ix][:, iy + 1] * (1 - tx) * ty + c[ix + 1][:, iy + 1] * tx * ty)
    terrain.height_field_raw += np.rint(z).astype(np.int16)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """a pyramid of the given slope with a flat top of `platform_size` metres"""
    x, y = np.arange(0, terrain.width), np.arange(0, terrain.length)
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    xx, yy = np.meshgrid(x, y, sparse=True)
    xx = ((cx - np.abs(cx - xx)) / cx).reshape(terrain.width, 1)
    yy = ((cy - np.abs(cy - yy)) / cy).reshape(1, terrain.length)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (max_height * xx * yy).astype(terrain.height_field_raw.dtype)
    ps = int(platform_size / terrain.horizontal_scale / 2)
    x1, x2, y1, y2 = terrain.width // 2 - ps, terrain.width // 2 + ps, terrain.length // 2 - ps, terrain.length // 2 + ps
    lo, hi = min(terrain.height_field_raw[x1, y1], 0), max(terrain.height_field_raw[x1, y1], 0)
    terrain.height_field_raw = np.clip(terrain.height_field_raw, lo, hi)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """concentric square steps rising (or, step_height < 0, descending) towards a platform in the middle"""
    step_width, step_height = int(step_width / terrain.horizontal_scale), int(step_height / terrain.vertical_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    height, x0, x1, y0, y1 = 0, 0, terrain.width, 0, terrain.length
    while (x1 - x0) > platform_size and (y1 - y0) > platform_size:
        x0 += step_width; x1 -= step_width; y0 += step_width; y1 -= step_width
        height += step_height
        terrain.height_field_raw[x0:x1, y0:y1] = height
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """`num_rects` random rectangles of random height, a flat platform kept free in the middle"""
    max_height = int(max_height / terrain.vertical_scale)
    min_size, max_size, platform_size = int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale), int(platform_size / terrain.horizontal_scale)
    (i, j) = terrain.height_field_raw.shape
    height_range = [-max_height, -max_height // 2, max_height // 2, max_height]
    width_range = range(min_size, max_size, 4)
    length_range = range(min_size, max_size, 4)
    for _ in range(num_rects):
        width, length = np.random.choice(width_range), np.random.choice(length_range)
        start_i, start_j = np.random.choice(range(0, i - width, 4)), np.random.choice(range(0, j - length, 4))
        terrain.height_field_raw[start_i:start_i + width, start_j:start_j + length] = np.random.choice(height_range)
    x1, x2, y1, y2 = (terrain.width - platform_size) // 2, (terrain.width + platform_size) // 2, (terrain.length - platform_size) // 2, (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """square stones of random height over a pit of `depth`, rows shifted at random; platform in the middle"""
    stone_size, stone_distance = int(stone_size / terrain.horizontal_scale), int(stone_distance / terrain.horizontal_scale)
    max_height, platform_size = int(max_height / terrain.vertical_scale), int(platform_size / terrain.horizontal_scale)
    height_range = np.arange(-max_height - 1, max_height, step=1)
    start_x, start_y = 0, 0
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    if terrain.length >= terrain.width:
        while start_y < terrain.length:
            stop_y = min(terrain.length, start_y + stone_size)
            start_x = np.random.randint(0, stone_size)
            stop_x = max(0, start_x - stone_distance)          # fill the first hole
            terrain.height_field_raw[0:stop_x, start_y:stop_y] = np.random.choice(height_range)
            while start_x < terrain.width:
                stop_x = min(terrain.width, start_x + stone_size)
                terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = np.random.choice(height_range)
                start_x += stone_size + stone_distance
            start_y += stone_size + stone_distance
    else:
        while start_x < terrain.width:
            stop_x = min(terrain.width, start_x + stone_size)
            start_y = np.random.randint(0, stone_size)
            stop_y = max(0, start_y - stone_distance)
            terrain.height_field_raw[start_x:stop_x, 0:stop_y] = np.random.choice(height_range)
            while start_y < terrain.length:
                stop_y = min(terrain.length, start_y + stone_size)
                terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = np.random.choice(height_range)
                start_y += stone_size + stone_distance
            start_x += stone_size + stone_distance
    x1, x2, y1, y2 = (terrain.width - platform_size) // 2, (terrain.width + platform_size) // 2, (terrain.length - platform_size) // 2, (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def gap_terrain(terrain, gap_size, platform_size=1.0):
    """a square moat of width `gap_size` around a platform (reference terrain.py:167-181)"""
    gap_size, platform_size = int(gap_size / terrain.horizontal_scale), int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1 = (terrain.length - platform_size) // 2
    x2 = x1 + gap_size
    y1 = (terrain.width - platform_size) // 2
    y2 = y1 + gap_size
    terrain.height_field_raw[cx - x2: cx + x2, cy - y2: cy + y2] = -1000
    terrain.height_field_raw[cx - x1: cx + x1, cy - y1: cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.0):
    """a square pit of `depth` in the middle (reference terrain.py:183-192)"""
    depth, platform_size = int(depth / terrain.vertical_scale), int(platform_size / terrain.horizontal_scale / 2)
    x1, x2, y1, y2 = terrain.length // 2 - platform_size, terrain.length // 2 + platform_size, terrain.width // 2 - platform_size, terrain.width // 2 + platform_size
    terrain.height_field_raw[x1:x2, y1:y2] = -depth


_GENERATORS = dict(random_uniform_terrain=random_uniform_terrain, pyramid_sloped_terrain=pyramid_sloped_terrain, pyramid_stairs_terrain=pyramid_stairs_terrain,
                   discrete_obstacles_terrain=discrete_obstacles_terrain, stepping_stones_terrain=stepping_stones_terrain, gap_terrain=gap_terrain, pit_terrain=pit_terrain)


class Terrain:
    def __init__(self, cfg, num_robots, num_agents=1):
        self.cfg = cfg
        self.num_robots, self.num_agents = num_robots, num_agents
        self.type = cfg.mesh_type
        self.env_info = None
        self._built = False
        if self.type in ["none", "plane"]:
            raise NotImplementedError("Terrain with mesh_type 'plane' / 'none' has no height field (legged_robot.py:966-967 creates a ground plane instead)")
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border

    def build(self):
        if self._built:
            return self
        cfg = self.cfg
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:                      # terrain.py:61-66: "selected" is a terrain CLASS name in this code base, so the
            self.curiculum()                    # per-type branch (:91-105) can only be reached with a dict-like terrain_kwargs
        elif isinstance(getattr(cfg, "terrain_kwargs", None), dict) and "type" in cfg.terrain_kwargs:
            self.selected_terrain()
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw
        self.heightfield_raw = self.height_field_raw
        self.env_origins = self.env_origins.astype(np.float32)
        self.agent_origins = np.repeat(self.env_origins[:, :, None, :], self.num_agents, axis=2)
        self.wall = np.zeros(self.height_field_raw.shape, bool)
        self.wall_sdf = np.full(self.height_field_raw.shape, 1e3, np.float32)
        self.wall_height, self.wall_top = 0.0, None
        self.ground_z = 0.0
        self.ground_height = self.height_field_raw.astype(np.float32) * np.float32(cfg.vertical_scale)
        self._built = True
        return self

    def add_terrain_to_sim(self, gym=None, sim=None, device="cpu"):
        self.device = device
        return self.build()

    def randomized_terrain(self):               # terrain.py:75-83
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):                        # terrain.py:85-92
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                self.add_terrain_to_map(self.make_terrain(j / self.cfg.num_cols + 0.001, i / self.cfg.num_rows), i, j)

    def selected_terrain(self):                 # terrain.py:94-108 (upstream reads self.vertical_scale, which does not exist: the cfg's are meant)
        kwargs = dict(self.cfg.terrain_kwargs)
        gen = _GENERATORS[kwargs.pop("type")]
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            t = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                           vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)
            gen(t, **kwargs.get("terrain_kwargs", kwargs))
            self.add_terrain_to_map(t, i, j)

    def make_terrain(self, choice, difficulty):   # terrain.py:110-145
        t = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                       vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        discrete_obstacles_height = 0.05 + difficulty * 0.2
        stepping_stones_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        gap_size = 1. * difficulty
        pit_depth = 1. * difficulty
        p = self.proportions
        if choice < p[0]:
            if choice < p[0] / 2:
                slope *= -1
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.)
        elif choice < p[1]:
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.)
            random_uniform_terrain(t, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[3]:
            if choice < p[2]:
                step_height *= -1
            pyramid_stairs_terrain(t, step_width=0.31, step_height=step_height, platform_size=3.)
        elif choice < p[4]:
            discrete_obstacles_terrain(t, discrete_obstacles_height, 1., 2., 20, platform_size=3.)
        elif len(p) > 5 and choice < p[5]:
            stepping_stones_terrain(t, stone_size=stepping_stones_size, stone_distance=stone_distance, max_height=0., platform_size=4.)
        elif len(p) > 6 and choice < p[6]:
            gap_terrain(t, gap_size=gap_size, platform_size=3.)
        else:
            pit_terrain(t, depth=pit_depth, platform_size=4.)
        return t

    def add_terrain_to_map(self, terrain, row, col):   # terrain.py:147-165
        i, j = row, col
        sx, ex = self.border + i * self.length_per_env_pixels, self.border + (i + 1) * self.length_per_env_pixels
        sy, ey = self.border + j * self.width_per_env_pixels, self.border + (j + 1) * self.width_per_env_pixels
        self.height_field_raw[sx:ex, sy:ey] = terrain.height_field_raw
        x1, x2 = int((self.env_length / 2. - 1) / terrain.horizontal_scale), int((self.env_length / 2. + 1) / terrain.horizontal_scale)
        y1, y2 = int((self.env_width / 2. - 1) / terrain.horizontal_scale), int((self.env_width / 2. + 1) / terrain.horizontal_scale)
        # The raster includes the border (this sub-terrain starts `border` pixels in) and the engine samples it from the world origin,
        # whereas upstream keeps origins border-free and shifts the MESH by -border_size (legged_robot.py:698-699,715-716): the same
        # scene in this engine's coordinates has the border added to the origins (robots spawn on their platform, not border_size off it).
        bs = self.border * terrain.horizontal_scale
        self.env_origins[i, j] = [bs + (i + 0.5) * self.env_length, bs + (j + 0.5) * self.env_width, np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale]
