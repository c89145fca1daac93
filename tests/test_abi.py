"""The C-ABI library loads on a CPU-only host and exports every entry point include/mqe_hip.h declares; the ctypes
mirror of the descriptor has the size the C compiler gives it.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

from mqe.engine import abi
from mqe.engine.hip_engine import LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "mqe_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(mqe_[a-z_0-9]+)\s*\(", h)))


def test_header_symbols_exported():
    assert os.path.isfile(LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mqe_hip.h but not exported by libmqe_hip.so"


def test_struct_mirror_size():
    lib = C.CDLL(LIB_PATH)
    assert lib.mqe_abi_version() == abi.ABI_VERSION
    assert lib.mqe_sizeof_desc() == C.sizeof(abi.SimDesc)
    import oracle_engine
    assert oracle_engine.load_library().mqo_sizeof_desc() == C.sizeof(abi.SimDesc)


def test_limits_agree_between_header_library_and_python_mirror():
    """VERDICT r5 "Next" 8: a limit, an export or a call's meaning changes -> MQE_ABI_VERSION changes; the three places that state the limits
    (include/mqe_hip.h, the built library through mqe_abi_limits, mqe/engine/abi.py) agree."""
    h = open(os.path.join(ROOT, "include", "mqe_hip.h")).read()
    hdr = {k: int(v) for k, v in re.findall(r"^#define (MQE_[A-Z_]+) (\d+)\b", h, re.M)}
    hc = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    enum = hc[hc.index("MQE_T_ROOT_STATE"):hc.index("MQE_T_COUNT")]
    t_count = len(set(re.findall(r"\bMQE_T_[A-Z_0-9]+\b", enum)))                       # enumerators before MQE_T_COUNT
    lib = C.CDLL(LIB_PATH)
    out = (C.c_int32 * 32)()
    n = lib.mqe_abi_limits(out, 32)
    names = ["MQE_ABI_VERSION", "MQE_MAX_AGENTS", "MQE_MAX_NPCS", "MQE_MAX_SPHERES", "MQE_MAX_PRIMS", "MQE_MAX_SELF_PAIRS", "MQE_MAX_LAYERS",
             "MQE_MAX_REWARD_TERMS", "MQE_NBODY", "MQE_NREP", "MQE_NDOF", "MQE_FRAME", "MQE_HIST"]
    assert n == len(names) + 1
    got = dict(zip(names, list(out)[:len(names)]))
    for k in names:
        assert got[k] == hdr[k], (k, got[k], hdr[k])
    assert out[len(names)] == abi.T_COUNT == t_count, (out[len(names)], abi.T_COUNT, t_count)
    py = dict(MQE_ABI_VERSION=abi.ABI_VERSION, MQE_MAX_AGENTS=abi.MAX_AGENTS, MQE_MAX_NPCS=abi.MAX_NPCS, MQE_MAX_SPHERES=abi.MAX_SPHERES, MQE_MAX_PRIMS=abi.MAX_PRIMS,
              MQE_MAX_SELF_PAIRS=abi.MAX_SELF_PAIRS, MQE_MAX_LAYERS=abi.MAX_LAYERS, MQE_MAX_REWARD_TERMS=abi.MAX_REWARD_TERMS, MQE_NBODY=abi.NBODY,
              MQE_NREP=abi.NREP, MQE_NDOF=abi.NDOF, MQE_FRAME=abi.FRAME, MQE_HIST=abi.HIST)
    assert py == {k: hdr[k] for k in py}
    assert lib.mqe_abi_version() == hdr["MQE_ABI_VERSION"] >= 15


def test_product_refuses_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from helpers import make_desc
    from mqe.engine.hip_engine import HipEngine
    d, k, _ = make_desc("go1gate", 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipEngine(d, k)


def test_bench_flop_basis_matches_the_kernel_constants():
    """bench.py prices k_gemm_h2 with the K it executes: MQE_HIST frames x MQE_H2_FRAME compact columns (csrc/mqe_common.hpp)."""
    import re
    src = open(os.path.join(ROOT, "multiagent-quadruped-environment_amd", "csrc", "mqe_common.hpp")).read()
    frame = int(re.search(r"#define MQE_H2_FRAME (\d+)", src).group(1))
    hist = int(re.search(r"#define MQE_HIST (\d+)", open(os.path.join(ROOT, "include", "mqe_hip.h")).read()).group(1))
    bench = open(os.path.join(ROOT, "bench.py")).read()
    k = re.search(r"^K_SPLIT = (\d+) \* (\d+)", bench, re.M)
    assert (int(k.group(1)), int(k.group(2))) == (hist, frame)


def test_bench_times_every_single_gpu_baseline_config():
    """BASELINE.json names five configs; the default bench.py run carries the headline (configs[1]) as `value` and the other ones that fit one
    GPU -- go1sheep-hard 2048, go1seesaw 4096, go1football-defender's 4096-env shard -- as `configs[]` (VERDICT r4 "Next" 2), at the sizes named."""
    import importlib.util
    import json
    import re
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    have = {(t, n) for (t, n, _, _) in bench.EXTRA_CONFIGS}
    for c in base[2:]:
        task = c.split(",")[0].strip()
        n = int(re.search(r"num_envs=(\d+)", c).group(1))
        n = 4096 if task == "go1football-defender" else n            # 32768 over 8 GPUs = 4096 per GPU
        assert (task, n) in have, (c, have)
    assert ("go1gate", 4096) in have and any(ev.get("MQE_COLLISION_MODEL") == "exact" for (_, _, ev, _) in bench.EXTRA_CONFIGS)
