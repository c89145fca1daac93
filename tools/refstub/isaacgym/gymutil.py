def parse_device_str(s):
    s = s.lower()
    if s in ("cpu", "cuda"):
        return s, 0
    t, i = s.split(":")
    return t, int(i)


def parse_sim_config(cfg, sim_params):
    for k, v in cfg.items():
        setattr(sim_params, k, v)


def parse_arguments(*a, **k):
    raise RuntimeError("isaacgym stub")


class AxesGeometry:
    pass


class WireframeSphereGeometry:
    def __init__(self, *a, **k):
        pass
