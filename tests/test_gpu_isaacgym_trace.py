"""Row H against the reference's own physics, HIP side (-m gpu): the Isaac Gym captures of tests/golden/ replayed through
mqe_simulate, one substep at a time from the recorded state and free-running, against the tolerances stated in
tests/isaacgym_replay.py; measured errors go to gpurun_out/isaacgym_replay_hip.jsonl.  Skips when no capture is present; the
synthetic file keeps the consumer exercised."""
import json
import os

import pytest

import isaacgym_replay as igr
from helpers import hip_engine, ROOT

pytestmark = pytest.mark.gpu


def test_hip_replays_a_synthetic_capture(tmp_path):
    """the oracle-written file through the HIP engine: one substep from the recorded state agrees to the HIP / oracle tolerance, the
    free run to trajectory-divergence level"""
    p = igr.synthetic_capture(str(tmp_path / "isaacgym_go1gate.npz"))
    res = igr.replay(hip_engine, p)
    assert res["one_step"]["base_pos"] < 2e-5 and res["one_step"]["joint_pos"] < 1e-4 and res["one_step"]["base_vel"] < 5e-3, res
    assert res["free_run"]["base_pos"] < 1e-3, res
    assert not igr.check(res)


@pytest.mark.skipif(not igr.captures(), reason="no Isaac Gym capture under tests/golden/ (tools/capture_isaacgym_trace.py needs an NVIDIA machine): row H stays parity-unpinned")
@pytest.mark.parametrize("path", igr.captures() or ["-"])
def test_hip_replays_the_isaacgym_capture(path):
    res = igr.replay(hip_engine, path)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "isaacgym_replay_hip.jsonl"), "a") as f:
            f.write(json.dumps(res) + "\n")
    print(json.dumps(res))
    bad = igr.check(res)
    assert not bad, f"{res['task']}: " + "; ".join(bad)
