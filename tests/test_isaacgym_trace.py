"""Row H against the reference's own physics, CPU side: every Isaac Gym capture under tests/golden/ (produced on an NVIDIA machine
by tools/capture_isaacgym_trace.py; none can be produced here) is replayed through the CPU oracle and held to the tolerances
stated in tests/isaacgym_replay.py.  Without a capture the consumer is still exercised on a synthetic file in the same format."""
import json
import os

import pytest

import isaacgym_replay as igr
from helpers import oracle_engine, ROOT


def test_replay_machinery_on_a_synthetic_capture(tmp_path):
    """format, joint-order permutation, one-step and free-running replays: a capture written by the oracle replays through the
    oracle with zero one-step error and zero drift (it pins nothing about Isaac Gym -- meta.source says what it is)"""
    p = igr.synthetic_capture(str(tmp_path / "isaacgym_go1gate.npz"))
    res = igr.replay(oracle_engine, p)
    assert res["source"].startswith("synthetic") and res["substeps"] == 24
    assert max(res["one_step"].values()) < 1e-6 and max(res["free_run"].values()) < 1e-6, res
    assert not igr.check(res)


@pytest.mark.skipif(not igr.captures(), reason="no Isaac Gym capture under tests/golden/ (tools/capture_isaacgym_trace.py needs an NVIDIA machine): row H stays parity-unpinned")
@pytest.mark.parametrize("path", igr.captures() or ["-"])
def test_oracle_replays_the_isaacgym_capture(path):
    res = igr.replay(oracle_engine, path)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "isaacgym_replay_oracle.jsonl"), "a") as f:
            f.write(json.dumps(res) + "\n")
    print(json.dumps(res))
    bad = igr.check(res)
    assert not bad, f"{res['task']}: " + "; ".join(bad)
