"""The two-half step of the env-sharded runner (include/mqe_hip.h: mqe_step_begin / mqe_step_end) and bench.py's sharded
path with the delayed all-gather, on the one GPU of the test box (two ranks on cuda:0 over gloo: the schedule and the
buffers are the ones of the RCCL run, the transport is not)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import make_desc, hip_engine
from mqe.engine import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_step_halves_equal_step():
    """mqe_step == mqe_step_begin; <host launches>; mqe_step_end -- bit for bit, with work of the host's own in between"""
    N = 64
    d0, k0, _ = make_desc("go1gate", N)
    d1, k1, _ = make_desc("go1gate", N)
    e0, e1 = hip_engine(d0, k0), hip_engine(d1, k1)
    e0.reset_all(); e1.reset_all()
    g = torch.Generator().manual_seed(77)
    scratch = torch.zeros(1 << 16, device="cuda")
    calls = []

    def between():
        calls.append(1)
        scratch.add_(1.0)          # a launch of the host's own between the policy and the physics kernels

    d2, k2, _ = make_desc("go1gate", N)
    e2 = hip_engine(d2, k2)            # and in three parts: mqe_step_head; <host>; mqe_step_tail; <host>; mqe_step_end
    e2.reset_all()
    for t in range(8):
        a = (torch.rand(N, 2, 3, generator=g) * 2 - 1).cuda().contiguous()
        e0.step(a)
        e1.step(a, between)
        e2.step(a, between, between)
    torch.cuda.synchronize()
    assert len(calls) == 24 and float(scratch[0]) == 24.0
    for kind in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_WRAPPER_OBS, abi.T_WRAPPER_REWARD, abi.T_RESET_BUF, abi.T_ACTIONS):
        assert torch.equal(e0.tensor(kind), e1.tensor(kind)), kind
        assert torch.equal(e0.tensor(kind), e2.tensor(kind)), kind


def test_packed_return_batch_is_obs_reward_done():
    """MQE_T_WRAPPER_PACKED = wrapper obs | reward | reset flags as N bytes (0/1), in the HIP engine and in the oracle: after
    reset_all (all flags 1) and along a rollout in which envs do reset (robots dropped from 3 m terminate on base contact)"""
    from helpers import oracle_engine
    N = 16
    d0, k0, _ = make_desc("go1gate", N)
    d1, k1, _ = make_desc("go1gate", N)
    eh, eo = hip_engine(d0, k0), oracle_engine(d1, k1)
    for e in (eh, eo):
        e.reset_all()
    seen = 0
    for t in range(40):
        a = torch.zeros(N, 2, 3)
        if t == 3:
            for e in (eh, eo):
                r = e.tensor(abi.T_ROOT_STATE)
                r[:4, 0, 2] += 3.0
        eh.step(a.cuda()); eo.step(a)
        torch.cuda.synchronize()
        for e in (eh, eo):
            o, r, f = e.tensor(abi.T_WRAPPER_OBS), e.tensor(abi.T_WRAPPER_REWARD), e.tensor(abi.T_RESET_BUF)
            pk = e.tensor(abi.T_WRAPPER_PACKED)
            assert pk.numel() == o.numel() + r.numel() + (N + 3) // 4
            assert torch.equal(pk[:o.numel()], o.reshape(-1)) and torch.equal(pk[o.numel():o.numel() + r.numel()], r.reshape(-1))
            assert torch.equal(pk[o.numel() + r.numel():].view(torch.uint8)[:N], f.reshape(-1).view(torch.uint8))
        seen += int(eh.tensor(abi.T_RESET_BUF).sum())
    assert seen > 0, "no env reset in this rollout: the done half of the check did not run"


def test_step_end_without_begin_is_an_error():
    d, k, _ = make_desc("go1gate", 4)
    e = hip_engine(d, k)
    e.reset_all()
    with pytest.raises(RuntimeError):
        e._call("step_end", e._stream())
    a = torch.zeros(4, 2, 3, device="cuda")
    e._call("step_begin", a.data_ptr(), e._stream())
    with pytest.raises(RuntimeError):          # a second begin before the end
        e._call("step_begin", a.data_ptr(), e._stream())
    e._call("step_end", e._stream())
    with pytest.raises(RuntimeError):          # tail without head
        e._call("step_tail", e._stream())
    e._call("step_head", a.data_ptr(), e._stream())
    with pytest.raises(RuntimeError):          # end before the tail
        e._call("step_end", e._stream())
    e._call("step_tail", e._stream())
    e._call("step_end", e._stream())
    torch.cuda.synchronize()


@pytest.mark.parametrize("task,port,world,extra", [
    ("go1gate", "29641", 2, []), ("go1football-defender", "29643", 2, ["--gather", "between"]),
    ("go1gate", "29645", 8, ["--gather", "between"]), ("go1gate", "29647", 8, ["--gather", "after"]), ("go1football-defender", "29649", 8, ["--gather", "after"]),
    ("go1gate", "29651", 8, ["--no_gather"]), ("go1gate", "29653", 8, ["--gather", "tail"]), ("go1football-defender", "29655", 2, ["--gather", "tail"])])
def test_bench_sharded_schedules_over_gloo(task, port, world, extra):
    """bench.py --gpus N as the driver launches it, all ranks on cuda:0 over gloo (world sizes 2 and 8 = the node the driver
    measures on): every step's batch is gathered and arrives whole -- bench.py asserts both -- under both schedules of the
    collective (issued between policy and physics of the next step | issued after the step, next step waits) and without a
    collective (per-GPU learners).  go1football-defender = BASELINE config 5's sharded path: 3 robots per env, the scripted
    defender and the reset draws keyed by the global env id."""
    env = dict(os.environ, MQE_BENCH_SELFTEST_GLOO="1", MASTER_ADDR="127.0.0.1")
    n = 128 if world == 2 else 32
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "12", "--warmup", "3",
           "--num_envs", str(n), "--no_cpu_baseline", "--task", task] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == world and r["steps"] == 12 and r["value"] > 0 and r["scaling"] == "weak"
    assert f"{n} per GPU ({n * world} total)" in r["config"]["workload"]
    if "--no_gather" in extra:
        assert r["collective"] == "none" and "no collective" in r["config"]["parallelism"]
    else:
        assert r["collective"]["schedule"] == (extra[1] if extra else "tail") and r["collective"]["gathers"] == 15


def test_rccl_process_group_options_single_rank():
    """the RCCL process group exactly as bench.py creates it for N > 1 (device_id, high-priority stream), world size 1: init,
    an asynchronous all_gather_into_tensor, wait, barrier, destroy"""
    code = r'''
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), pg_options=opts)
torch.cuda.set_device(0)
x = torch.arange(1024, device="cuda", dtype=torch.float32).reshape(32, 32)
y = torch.empty(32, 32, device="cuda")
w = dist.all_gather_into_tensor(y, x, async_op=True)
w.wait()
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(x, y)
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl ok")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_return_buffer_receives_the_step_outputs():
    """mqe_set_return_buffer: the step writes obs | reward | done into the caller's tensor (what the wrappers hand out as the
    step's fresh tensors) and leaves the engine's own buffer alone; NULL switches back"""
    N = 32
    d0, k0, _ = make_desc("go1gate", N)
    d1, k1, _ = make_desc("go1gate", N)
    e0, e1 = hip_engine(d0, k0), hip_engine(d1, k1)
    e0.reset_all(); e1.reset_all()
    g = torch.Generator().manual_seed(3)
    own_before = e1.tensor(abi.T_WRAPPER_PACKED).clone()
    bufs = []
    for t in range(5):
        a = (torch.rand(N, 2, 3, generator=g) * 2 - 1).cuda().contiguous()
        e0.step(a)
        buf = torch.full_like(e1.tensor(abi.T_WRAPPER_PACKED), float("nan"))
        e1.set_return_buffer(buf)
        e1.step(a)
        e1.set_return_buffer(None)
        torch.cuda.synchronize()
        assert torch.equal(buf.view(torch.int32), e0.tensor(abi.T_WRAPPER_PACKED).view(torch.int32)), t      # bit patterns: the tail is bytes
        bufs.append(buf)
    assert torch.equal(e1.tensor(abi.T_WRAPPER_PACKED).view(torch.int32), own_before.view(torch.int32))
    assert not torch.equal(bufs[0].view(torch.int32), bufs[-1].view(torch.int32))
    a = torch.zeros(N, 2, 3, device="cuda")
    e0.step(a); e1.step(a)                     # back on the engine's own buffer
    torch.cuda.synchronize()
    assert torch.equal(e1.tensor(abi.T_WRAPPER_PACKED).view(torch.int32), e0.tensor(abi.T_WRAPPER_PACKED).view(torch.int32))


def test_full_size_batch_is_the_union_of_its_shards(monkeypatch):
    """BASELINE config 1 (4096 envs x 2 agents) against small batches that the oracle tests cover: envs [g0, g0+32) of the full
    batch and the same global env ids run alone (env_id_offset) go through identical kernels row by row / wave by wave, so 20
    fused steps must agree BIT FOR BIT -- state, returned batch, reset flags (the layer-0 kernel is pinned to the split-f16
    one, which the small batch would not pick by itself)."""
    monkeypatch.setenv("MQE_GEMM_SPLIT", "1")
    NF, NS = 4096, 32
    df, kf, _ = make_desc("go1gate", NF)
    ef = hip_engine(df, kf)
    ef.reset_all()
    small = []
    for g0 in (0, 2016, NF - NS):
        d, k, _ = make_desc("go1gate", NS, env_id_offset=g0)
        e = hip_engine(d, k)
        e.reset_all()
        small.append((g0, e))
    g = torch.Generator().manual_seed(11)
    kinds = (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_WRAPPER_OBS, abi.T_WRAPPER_REWARD, abi.T_RESET_BUF, abi.T_EPISODE_LENGTH)
    for t in range(-1, 20):
        if t >= 0:
            a = torch.rand(NF, 2, 3, generator=g) * 2 - 1
            ef.step(a.cuda().contiguous())
            for g0, e in small:
                e.step(a[g0:g0 + NS].cuda().contiguous())
        torch.cuda.synchronize()
        for g0, e in small:
            for kind in kinds:
                full = ef.tensor(kind)
                per = full.shape[0] // NF
                assert torch.equal(full[g0 * per:(g0 + NS) * per], e.tensor(kind)), f"step {t}, envs from {g0}, tensor kind {kind}"
