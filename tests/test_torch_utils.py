"""isaacgym.torch_utils is external to the reference, so the restatement is pinned by analytic identities."""
import math

import torch

from mqe.utils import torch_utils as tu


def rand_quat(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def test_rotate_inverse_roundtrip_and_norm():
    q, v = rand_quat(64), torch.randn(64, 3)
    w = tu.quat_rotate_inverse(q, tu.quat_rotate(q, v))
    assert torch.allclose(w, v, atol=1e-5)
    assert torch.allclose(tu.quat_rotate(q, v).norm(dim=-1), v.norm(dim=-1), atol=1e-5)
    assert torch.allclose(tu.quat_apply(q, v), tu.quat_rotate(q, v), atol=1e-5)


def test_known_rotation_xyzw():
    # 90 deg about z (xyzw): x axis -> y axis; inverse maps y -> x
    s = math.sqrt(0.5)
    q = torch.tensor([[0.0, 0.0, s, s]])
    assert torch.allclose(tu.quat_rotate(q, torch.tensor([[1.0, 0, 0]])), torch.tensor([[0.0, 1, 0]]), atol=1e-6)
    assert torch.allclose(tu.quat_rotate_inverse(q, torch.tensor([[0.0, 1, 0]])), torch.tensor([[1.0, 0, 0]]), atol=1e-6)
    # projected gravity of an upright base
    assert torch.allclose(tu.quat_rotate_inverse(torch.tensor([[0.0, 0, 0, 1]]), torch.tensor([[0.0, 0, -1]])), torch.tensor([[0.0, 0, -1]]))


def test_euler_roundtrip_and_range():
    g = torch.Generator().manual_seed(1)
    r = (torch.rand(128, generator=g) - 0.5) * 2.0
    p = (torch.rand(128, generator=g) - 0.5) * 2.0
    y = (torch.rand(128, generator=g) - 0.5) * 6.0
    q = tu.quat_from_euler_xyz(r, p, y)
    assert torch.allclose(q.norm(dim=-1), torch.ones(128), atol=1e-6)
    rr, pp, yy = tu.get_euler_xyz(q)
    two_pi = 2 * math.pi
    for a, b in ((rr, r), (pp, p), (yy, y)):
        assert ((a >= 0) & (a < two_pi + 1e-6)).all()
        d = (a - b + math.pi) % two_pi - math.pi
        assert d.abs().max() < 1e-4
    # the reference's termination code maps (pi, 2pi) back to negatives (legged_robot_field.py:126-127)
    rr2 = rr.clone()
    rr2[rr2 > math.pi] -= two_pi
    assert torch.allclose(rr2, r, atol=1e-4)


def test_quat_mul_conjugate_identity():
    q = rand_quat(32, 3)
    e = tu.quat_mul(q, tu.quat_conjugate(q))
    assert torch.allclose(e, torch.tensor([0.0, 0, 0, 1]).expand(32, 4), atol=1e-6)
