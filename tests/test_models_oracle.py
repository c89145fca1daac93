"""MLPs of the path (rows C, D, F) and the URDF-derived model, CPU oracle vs the reference's golden vectors."""
import ctypes as C

import numpy as np
import torch

from helpers import golden, make_desc, oracle_engine
from mqe.utils import urdf_model


def test_actuator_net_matches_torchscript():
    z = golden("actuator_net")
    d, k, _ = make_desc("go1gate", 1)
    e = oracle_engine(d, k)
    f = e.lib.mqo_actuator_net
    f.argtypes, f.restype = [C.c_void_p, C.c_void_p], C.c_float
    for x, y in ((z["x"], z["y"]), (z["spot_x"], z["spot_y"])):
        rows = [np.ascontiguousarray(r, np.float32) for r in x]          # keep alive across the foreign calls
        got = np.array([f(e.h, r.ctypes.data) for r in rows])
        np.testing.assert_allclose(got, y, rtol=2e-5, atol=2e-5)
    assert abs(z["spot_y"][0] - 19.5816) < 1e-3          # SURVEY 8(c) spot value


def test_adaptation_module_matches_torchscript():
    z = golden("adaptation_module")
    d, k, _ = make_desc("go1gate", 1)
    e = oracle_engine(d, k)
    f = e.lib.mqo_policy_forward
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lat, act = np.zeros(2, np.float32), np.zeros(12, np.float32)
    for i in range(z["x"].shape[0]):
        xi = np.ascontiguousarray(z["x"][i], np.float32)
        f(e.h, xi.ctypes.data, lat.ctypes.data, act.ctypes.data)
        np.testing.assert_allclose(lat, z["y"][i], rtol=1e-4, atol=2e-5)
    zero = np.zeros(2100, np.float32)
    f(e.h, zero.ctypes.data, lat.ctypes.data, act.ctypes.data)
    np.testing.assert_allclose(lat, [7.8227, 0.8373], atol=1e-3)   # f(0) known answer


def test_go1_model_from_urdf_numbers():
    m = urdf_model.load_model("go1")
    assert m["dof_names"][:3] == ["FL_hip_joint", "FL_thigh_joint", "FL_calf_joint"] and m["dof_names"][9] == "RR_hip_joint"
    assert len(m["reported_body_names"]) == 17 and m["reported_body_names"][4] == "FL_foot"
    # trunk 4.8 + imu 0.001; hip .510299; thigh .898919; calf .158015 + foot .06  (go1.urdf inertial blocks)
    np.testing.assert_allclose(m["mass"][0], 4.801, atol=1e-9)
    np.testing.assert_allclose(m["mass"][3], 0.158015 + 0.06, atol=1e-9)
    np.testing.assert_allclose(m["total_mass"], 4.801 + 4 * (0.510299 + 0.898919 + 0.218015), atol=1e-9)
    assert m["parent"] == [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11]
    assert len(m["sphere_body"]) == 27 and m["sphere_radius"][0] == 0.02
    for I in m["inertia"]:
        assert np.all(np.linalg.eigvalsh(np.asarray(I)) > 0)


def scripted_body(tmp_path, hidden=(128, 64), seed=5):
    """A TorchScript body as upstream ships it (Sequential of Linear / ELU, go1.py:397) with hidden sizes that are NOT the
    stand-in's 512-256-128: written to <tmp>/body_latest.jit, read back through the product loader (policy_weights.load_body)."""
    from mqe.utils import policy_weights as pw
    torch.manual_seed(seed)
    dims = (pw.BODY_IN,) + tuple(hidden) + (pw.BODY_OUT,)
    layers = []
    for i in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.mul_(2.0 if i else 1.0)
            lin.bias.uniform_(-0.3, 0.3)
        layers.append(lin)
        if i < len(dims) - 2:
            layers.append(torch.nn.ELU())
    net = torch.nn.Sequential(*layers).eval()
    torch.jit.script(net).save(str(tmp_path / "body_latest.jit"))
    Ws, bs, synthetic = pw.load_body(str(tmp_path))
    assert synthetic is False and [w.shape for w in Ws] == [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
    for w, b, lin in zip(Ws, bs, [l for l in layers if isinstance(l, torch.nn.Linear)]):
        assert np.array_equal(w, lin.weight.detach().numpy()) and np.array_equal(b, lin.bias.detach().numpy())
    return net, Ws, bs


def torch_adaptation():
    from mqe.utils import policy_weights as pw
    Ws, bs = pw.load_adaptation_module()
    layers = []
    for i, (w, b) in enumerate(zip(Ws, bs)):
        lin = torch.nn.Linear(w.shape[1], w.shape[0])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w)); lin.bias.copy_(torch.from_numpy(b))
        layers.append(lin)
        if i < len(Ws) - 1:
            layers.append(torch.nn.ELU())
    return torch.nn.Sequential(*layers).eval()


def test_real_body_file_of_another_shape_runs_through_the_loader(tmp_path):
    """Row D: body_latest.jit is missing upstream, so what can be pinned is the path a real file takes: TorchScript ->
    load_body -> descriptor -> oracle forward == the TorchScript module's own output on the same inputs (plain torch fp32),
    for a network whose depth and widths differ from the synthetic stand-in (2102-128-64-12)."""
    net, Ws, bs = scripted_body(tmp_path)
    ada = torch_adaptation()
    d, k, _ = make_desc("go1gate", 1, body=(Ws, bs))
    assert [d.body.dims[i] for i in range(d.body.n_layers + 1)] == [2102, 128, 64, 12]
    e = oracle_engine(d, k)
    f = e.lib.mqo_policy_forward
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    g = torch.Generator().manual_seed(2)
    lat, act = np.zeros(2, np.float32), np.zeros(12, np.float32)
    for i in range(8):
        x = torch.randn(2100, generator=g) * (0.3 if i else 0.0)
        with torch.no_grad():
            lt = ada(x[None])
            at = net(torch.cat([x[None], lt], dim=1))
        xi = np.ascontiguousarray(x.numpy(), np.float32)
        f(e.h, xi.ctypes.data, lat.ctypes.data, act.ctypes.data)
        np.testing.assert_allclose(lat, lt[0].numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(act, at[0].numpy(), rtol=1e-4, atol=5e-5)
