"""Manifold reduction of a robot's one-sided contacts (round 6, VERDICT r5 "Next" 4; desc.edge_contacts bit 8, off by default): a robot that touches the static world at more points than
its eight slots keeps every contact that penetrates by more than 1 mm first (deepest 2 mm class first), then feature order -- instead of the first eight in
feature order whatever their depth.
Counted in MQE_T_CONTACT_REDUCED; MQE_T_CONTACT_OVERFLOW stays zero for it.  The specification is oracle/mqe_oracle.c ("manifold reduction");
the HIP engine ranks the wavefront's candidate lanes (kernels_physics.hpp) and must produce the same list."""
import numpy as np
import pytest
import torch

from helpers import make_desc, oracle_engine, hip_engine, close
from mqe.engine import abi


def _fallen(e, N, roll=1.5707963, z=0.10, tilt=0.0):
    """robot 0 of every env lying on its side (rolled about x, optionally tilted about y so that the touching points differ in depth), robot 1 standing"""
    root = e.tensor(abi.T_ROOT_STATE)
    r = root.cpu().clone() if root.is_cuda else root.clone()
    cr, sr, cp, sp = np.cos(roll / 2), np.sin(roll / 2), np.cos(tilt / 2), np.sin(tilt / 2)
    # q = q_y(tilt) * q_x(roll), xyzw
    q = torch.tensor([cp * sr, sp * cr, -sp * sr, cp * cr], dtype=torch.float32)
    r[:, 0, 2] = z
    r[:, 0, 3:7] = q
    r[:, 0, 7:] = 0.0
    root.copy_(r.to(root.device))


@pytest.mark.parametrize("model", ["capsule", "exact"])
def test_a_fallen_robot_keeps_its_deepest_contacts(model):
    N = 3
    d, k, _ = make_desc("go1gate", N, collision_model=model, edge_contacts=3 | 8)
    e = oracle_engine(d, k)
    e.reset_all()
    _fallen(e, N, tilt=0.12)
    _, _, c = e.debug_dynamics(0, 0)
    mine = c[(c[:, 0] == 0) & (c[:, 2] < 0)]
    assert len(mine) == 8                                            # at the cap ...
    e.simulate()
    assert int(e.tensor(abi.T_CONTACT_OVERFLOW).sum()) == 0          # ... by reduction, not by truncation
    assert int(e.tensor(abi.T_CONTACT_REDUCED).sum()) == N
    assert mine[:, 4].min() <= -0.02 and mine[:, 4].max() < d.contact_offset      # lying 10 cm above the slab's top with a 12 cm half width: deep contacts exist


def test_reduction_prefers_depth_over_feature_order():
    """a robot lying on its side, tilted so that one end is deeper: feature order would keep the same feet / knees either way; the reduction keeps
    the features of the lower end."""
    N = 2
    d, k, _ = make_desc("go1gate", N, collision_model="capsule", edge_contacts=3 | 8)
    e = oracle_engine(d, k)
    e.reset_all()
    _fallen(e, N, tilt=0.25)             # rotated about +y: the front end is lower
    root = e.tensor(abi.T_ROOT_STATE)[0, 0].clone()
    _, _, c = e.debug_dynamics(0, 0)
    mine = c[(c[:, 0] == 0) & (c[:, 2] < 0)]
    assert len(mine) == 8
    e2 = oracle_engine(*make_desc("go1gate", N, collision_model="capsule", edge_contacts=3 | 8)[:2])
    e2.reset_all()
    _fallen(e2, N, tilt=-0.25)           # the rear end is lower
    _, _, c2 = e2.debug_dynamics(0, 0)
    mine2 = c2[(c2[:, 0] == 0) & (c2[:, 2] < 0)]
    # link ids of the kept contacts (column 1: body index; 1-6 front legs, 7-12 rear legs): more rear links when the rear is deeper, and vice versa
    rear = lambda m_: int(((m_[:, 1] >= 7)).sum()); front = lambda m_: int(((m_[:, 1] >= 1) & (m_[:, 1] <= 6)).sum())
    assert front(mine) > front(mine2) and rear(mine2) > rear(mine), (mine[:, 1], mine2[:, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["capsule", "exact"])
def test_reduction_hip_matches_the_specification(model):
    N = 6
    d1, k1, _ = make_desc("go1gate", N, collision_model=model, edge_contacts=3 | 8)
    d2, k2, _ = make_desc("go1gate", N, collision_model=model, edge_contacts=3 | 8)
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    for tilt in (0.12, -0.2, 0.3):
        _fallen(eo, N, tilt=tilt)
        eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda())
        eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
        torch.cuda.synchronize()
        for env in (0, N - 1):
            _, ch = eh.debug_dynamics(env, 0)
            _, _, co = eo.debug_dynamics(env, 0)
            assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (tilt, env, ch[:, :4], co[:, :4])
            close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")
    r0h, r0o = int(eh.tensor(abi.T_CONTACT_REDUCED).sum()), int(eo.tensor(abi.T_CONTACT_REDUCED).sum())
    eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    assert int(eh.tensor(abi.T_CONTACT_REDUCED).sum()) - r0h == int(eo.tensor(abi.T_CONTACT_REDUCED).sum()) - r0o == N
    assert int(eh.tensor(abi.T_CONTACT_OVERFLOW).sum()) == int(eo.tensor(abi.T_CONTACT_OVERFLOW).sum()) == 0
    close(eh.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_ROOT_STATE), atol=2e-5, what="root state one substep after the reduction")
