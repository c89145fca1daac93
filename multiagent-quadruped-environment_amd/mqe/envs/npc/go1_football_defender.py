"""Two learners + a scripted third robot + ball, reference mqe/envs/npc/go1_football_defender.py:12-80.  The
defender's command (_get_defender_action) is computed in the engine from the current state."""
from mqe.envs.go1.go1 import Go1


class Go1FootballDefender(Go1):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.npc_collision, self.fix_npc_base_link, self.npc_gravity = True, False, True
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
