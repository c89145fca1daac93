import pytest

from helpers import oracle_engine
from wrapper_replay import wrapper_replay


@pytest.mark.parametrize("name", ["gate", "sheep_hard", "sheep_easy", "seesaw", "football_defender", "pushbox", "rotation", "bridge", "wrestling", "tug"])
def test_oracle_wrappers_match_reference(name):
    assert wrapper_replay(name, oracle_engine)
