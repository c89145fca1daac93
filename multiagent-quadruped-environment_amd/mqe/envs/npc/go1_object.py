"""Go1 + one passive object (ball / seesaw ...), reference mqe/envs/npc/go1_object.py:14-62.  The object is part of
the engine's scene description (`npc_kind`), so the subclass only carries the asset flags."""
from mqe.envs.go1.go1 import Go1


class Go1Object(Go1):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.npc_collision = getattr(cfg.asset, "npc_collision", True)
        self.fix_npc_base_link = getattr(cfg.asset, "fix_npc_base_link", False)
        self.npc_gravity = getattr(cfg.asset, "npc_gravity", True)
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
