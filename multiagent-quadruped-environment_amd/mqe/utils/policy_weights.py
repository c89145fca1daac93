"""Weights of the three MLPs on the step() hot path, as plain numpy arrays.

* actuator net  6-32-32-1 softsign   (reference: resources/actuator_nets/unitree_go1.pt, loaded go1.py:367)
* adaptation module 2100-256-128-2 ELU (reference: .../walk_these_ways/adaptation_module_latest.jit, go1.py:398)
* body 2102-...-12 ELU (reference: .../walk_these_ways/body_latest.jit, go1.py:397) -- the file is listed in the
  reference's .MISSING_LARGE_BLOBS, so only its I/O contract (2102 in, 12 out) is pinned.  When no real file is
  given we build a deterministic stand-in (512-256-128 hidden, ELU): every number produced with it is labelled
  "synthetic body" in bench/test output.

Weight layout everywhere in this package: W[l] has shape (out, in) (torch.nn.Linear convention), b[l] (out,).
"""
import os

import numpy as np

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")

BODY_HIDDEN = (512, 256, 128)
BODY_IN = 2102
BODY_OUT = 12


def synthetic_body(seed: int = 0, hidden=BODY_HIDDEN, out_gain: float = 0.5):
    """Deterministic stand-in for body_latest.jit: N(0, 1/fan_in) weights, zero bias, small output gain."""
    rs = np.random.RandomState(seed)
    dims = (BODY_IN,) + tuple(hidden) + (BODY_OUT,)
    Ws, bs = [], []
    for li in range(len(dims) - 1):
        fan_in, fan_out = dims[li], dims[li + 1]
        g = out_gain if li == len(dims) - 2 else 1.0
        Ws.append((rs.standard_normal((fan_out, fan_in)) * (g / np.sqrt(fan_in))).astype(np.float32))
        bs.append(np.zeros((fan_out,), np.float32))
    return Ws, bs


def _load_npz_mlp(path):
    z = np.load(path)
    n = len([k for k in z.files if k.startswith("W")])
    return [z[f"W{i}"].astype(np.float32) for i in range(n)], [z[f"b{i}"].astype(np.float32) for i in range(n)]


def load_actuator_net(path=None):
    return _load_npz_mlp(path or os.path.join(ASSET_DIR, "actuator_net_unitree_go1.npz"))


def load_adaptation_module(path=None):
    return _load_npz_mlp(path or os.path.join(ASSET_DIR, "adaptation_module.npz"))


def load_torchscript_mlp(path):
    """Load a Sequential(Linear, act, Linear, ...) TorchScript file (e.g. a real body_latest.jit)."""
    import torch

    m = torch.jit.load(path, map_location="cpu")
    sd = m.state_dict()
    keys = sorted({int(k.split(".")[0]) for k in sd})
    return ([sd[f"{k}.weight"].numpy().astype(np.float32) for k in keys],
            [sd[f"{k}.bias"].numpy().astype(np.float32) for k in keys])


def load_body(policy_dir=None, seed: int = 0):
    """(Ws, bs, is_synthetic).  Uses <policy_dir>/body_latest.jit when it exists."""
    if policy_dir:
        p = os.path.join(policy_dir, "body_latest.jit")
        if os.path.isfile(p):
            Ws, bs = load_torchscript_mlp(p)
            assert Ws[0].shape[1] == BODY_IN and Ws[-1].shape[0] == BODY_OUT, "body I/O must be 2102 -> 12 (go1.py:404,29)"
            return Ws, bs, False
    Ws, bs = synthetic_body(seed)
    return Ws, bs, True
