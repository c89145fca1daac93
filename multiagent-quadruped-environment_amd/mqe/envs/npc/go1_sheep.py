"""Go1 + scripted flocking sheep, reference mqe/envs/npc/go1_sheep.py:21-64.  The flocking rule (_step_npc) runs in
the engine's post-physics kernel; `sheep_pos_avg` / `sheep_pos_var` are engine tensors exposed by Go1."""
from mqe.envs.go1.go1 import Go1


class Go1Sheep(Go1):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.npc_collision, self.fix_npc_base_link, self.npc_gravity = True, False, True
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.sheep_movement_scale = cfg.asset.sheep_movement_scale
        self.sheep_movement_randomness = cfg.asset.sheep_movement_randomness
        self.sheep_movement_range = cfg.asset.sheep_movement_range
