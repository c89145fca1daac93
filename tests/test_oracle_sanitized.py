"""SURVEY 5 (memory / undefined-behaviour detection): the CPU oracle rebuilt with -fsanitize=address,undefined (oracle/Makefile target
`asan`) runs fused steps of one scene per code path -- robots only, free NPCs (flock), the 1-dof link, the free box, scenery, three
robots + ball -- in a subprocess with the asan runtime preloaded; any out-of-bounds access, use of uninitialised stack through UB,
signed overflow or misaligned access aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import torch
from helpers import make_desc, oracle_engine
from mqe.engine import abi
for task in ("go1gate", "go1sheep-hard", "go1seesaw", "go1pushbox", "go1bridge", "go1football-defender", "go1tug"):
    d, k, _ = make_desc(task, 3)
    e = oracle_engine(d, k)
    e.lib.mqo_set_num_threads(1)
    e.reset_all()
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator().manual_seed(2)
    r = e.tensor(abi.T_ROOT_STATE)
    r[0, 0, 2] += 0.4                      # one robot dropped: trunk / leg contacts, resets
    for t in range(6):
        e.step(torch.rand(3, Aw, 3, generator=g) * 2 - 1)
    assert torch.isfinite(e.tensor(abi.T_ROOT_STATE)).all()
    e.close()
print("sanitized run ok")
'''


def test_oracle_runs_clean_under_asan_and_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isfile(libasan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, MQE_ORACLE_LIB="libmqe_oracle_asan.so", OMP_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, r.stdout[-1500:] + r.stderr[-6000:]
