"""The CPU oracle replays the reference's own step() traces (everything around the physics)."""
import pytest

from helpers import oracle_engine
from replay import replay


@pytest.mark.parametrize("name", ["gate", "seesaw", "football", "sheep", "football1v1", "football2v2", "pushbox", "rotation", "bridge", "wrestling", "tug"])
def test_oracle_matches_reference_trace(name):
    assert replay(name, oracle_engine)
