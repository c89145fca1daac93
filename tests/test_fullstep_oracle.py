"""The CPU oracle replays the reference's own step() traces (everything around the physics)."""
import pytest

from helpers import oracle_engine
from replay import replay


@pytest.mark.parametrize("name", ["gate", "seesaw", "football", "sheep", "football1v1", "football2v2", "pushbox", "rotation", "bridge", "wrestling", "tug", "gate_cmd", "pushbox_curriculum"])
def test_oracle_matches_reference_trace(name):
    assert replay(name, oracle_engine)


@pytest.mark.parametrize("ctrl", ["P", "V", "T"])
def test_oracle_matches_reference_trace_low_level_control(ctrl):
    """control types P / V / T (Go1.step's else-branch): joint-space actions, PD / velocity / torque law"""
    from replay import replay_joint
    assert replay_joint(ctrl, oracle_engine)
