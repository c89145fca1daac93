"""TerrainPerlin: the whole map is fractal gradient noise, no tracks, no walls (reference mqe/utils/terrain/perlin.py:9-32,88-117;
registry entry mqe/utils/terrain/__init__.py:6).  Selectable through `cfg.terrain.selected = "TerrainPerlin"`.

Upstream hands the int16 height samples to PhysX as a triangle mesh; here they reach the engine as the relief of the walkable
surface (`ground_height` [m] at the raster points (i hs, j hs), the mesh's vertices), next to an empty wall set.  Quirks kept:
rows / columns are swapped in the sample counts (tot_cols from the x size, tot_rows from the y size) and asserted consistent
(:14-18), so the map has to be square; env origins sit at the centre of each (terrain_length x terrain_width) cell at the height
sample under them (:104-117); there are no per-agent origins (upstream's LeggedRobot then reuses the env origins,
legged_robot.py:987-990 -- which only works for one agent per env; here every agent of an env gets the env's origin and the
start states of cfg.init_state spread them)."""
import numpy as np

from .barrier_track import fractal_noise


class TerrainPerlin:
    def __init__(self, cfg, num_envs, num_agents=1):
        self.cfg = cfg
        self.num_envs, self.num_agents = num_envs, num_agents
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.xSize = cfg.terrain_length * cfg.num_rows
        self.ySize = cfg.terrain_width * cfg.num_cols
        self.tot_cols = int(self.xSize / cfg.horizontal_scale)
        self.tot_rows = int(self.ySize / cfg.horizontal_scale)
        assert self.xSize == cfg.horizontal_scale * self.tot_rows and self.ySize == cfg.horizontal_scale * self.tot_cols
        self.env_info = None
        self._built = False

    def build(self):
        if self._built:
            return self
        cfg = self.cfg
        self.heightsamples_float = fractal_noise(self.xSize, self.ySize, self.tot_rows, self.tot_cols, **cfg.TerrainPerlin_kwargs)
        self.heightsamples = (self.heightsamples_float * (1 / cfg.vertical_scale)).astype(np.int16)
        self.heightfield_raw = self.heightsamples
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3), np.float32)
        for r in range(cfg.num_rows):
            for c in range(cfg.num_cols):
                ox, oy = (r + 0.5) * self.env_length, (c + 0.5) * self.env_width
                self.env_origins[r, c] = [ox, oy, self.heightsamples[int(ox / cfg.horizontal_scale), int(oy / cfg.horizontal_scale)] * cfg.vertical_scale]
        self.agent_origins = np.repeat(self.env_origins[:, :, None, :], self.num_agents, axis=2)
        # the engine's view: no wall set, the relief is the ground
        self.wall = np.zeros(self.heightsamples.shape, bool)
        self.wall_sdf = np.full(self.heightsamples.shape, 1e3, np.float32)
        self.wall_height, self.wall_top = 0.0, None
        self.ground_z = 0.0
        self.ground_height = (self.heightsamples.astype(np.float32) * np.float32(cfg.vertical_scale))
        self._built = True
        return self

    def add_terrain_to_sim(self, gym=None, sim=None, device="cpu"):
        """name kept for source compatibility (reference perlin.py:95); gym / sim are ignored"""
        self.device = device
        return self.build()
