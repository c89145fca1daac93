#!/usr/bin/env python3
"""bench.py -- env-steps/s of the go1gate step() hot path (BASELINE.json metric) on N MI355X GPUs.

  python bench.py --gpus 1 --steps 500 --warmup 50
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one wrapper-level env.step(): task action scaling -> locomotion policy (adaptation + body MLP over the
30-frame history) -> 4 x (actuator-net torques + 5 ms of rigid-body dynamics with contact) -> post-physics step
(termination, in-kernel resets, observations) -> task observation/reward, for `num_envs` envs x 2 agents with
fresh U(-1,1) actions every step (generator seed 1234).  Environments shard across ranks (weak scaling: 4096 envs
per GPU); the only collective is the RCCL all-gather of the returned (obs, reward, done) batch.

Prints ONE JSON line (rank 0).  value = agents x envs(all ranks) x steps / max-over-ranks wall time.
`roofline`: dominant kernel (largest share of GPU time, HIP events on the launch stream over the timed region) charged with the
whole step's algorithmic bytes; `roofline_per_kernel`: every kernel class against ITS OWN bytes / flops;
`strict_f32`: the same workload with every policy GEMM on the exact-f32 kernels (the companion of `dtype: f32`);
`cpu_baseline` / `cpu_baseline_n4`: the build's CPU restatement (oracle/, kind "port") timed on this box's host cores at the
GPU run's batch size (8 steps) and at N = 4 (BASELINE config 1).
NOTE: body_latest.jit is missing from the reference snapshot, so the locomotion-policy body is the deterministic
synthetic 2102-512-256-128-12 ELU MLP (mqe/utils/policy_weights.py); the adaptation module and actuator net are real.
"""
import argparse
import contextlib
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multiagent-quadruped-environment_amd"))

import torch  # noqa: E402

PROF_NAMES = ["policy_layer0(k_gemm_h2 | k_gemm_f32)", "policy_tail(k_policy_tail | k_gemm_f32 x5 + k_body_l0_finish + k_post_policy)",
              "torques(unfused path only)", "substeps(k_substeps: 4 x {actuator-net MFMA + physics substep})",
              "post(k_post_physics + k_reset_history)", "misc(k_wrapper_command + k_pre_policy)"]
PROF_KERNEL = ["k_gemm_h2", "k_gemm_f32", "k_compute_torques_mfma", "k_substeps", "k_post_physics", "k_pre_policy"]
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak
K_SPLIT = 30 * 48               # K of k_gemm_h2 over the history: MQE_HIST frames x MQE_H2_FRAME compact columns (csrc/mqe_common.hpp; tests/test_abi.py keeps the two in step)


def abi_mod():
    from mqe.engine import abi
    return abi


def make_args(task, num_envs, seed, device):
    from mqe.utils.helpers import finish_args
    a = types.SimpleNamespace(task=task, num_envs=num_envs, seed=seed, headless=True, record_video=False,
                              sim_device=device, pipeline="gpu", subscenes=0, num_threads=0)
    return finish_args(a)


def binary_info():
    """Which libmqe_hip.so ran: __graft_entry__.build_engine() recompiles only when a source is newer than the library, so a GPU box
    normally loads the library that was cross-compiled in the build container and shipped with the snapshot; the sidecar written by
    the build says where and when it was made."""
    import hashlib
    import socket
    so = os.path.join(ROOT, "multiagent-quadruped-environment_amd", "csrc", "libmqe_hip.so")
    so = os.environ.get("MQE_HIP_LIB", so)
    info = {"path": os.path.relpath(so, ROOT), "so_mtime": None, "so_sha16": None, "built_on_box": None}
    try:
        info["so_mtime"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(os.path.getmtime(so)))
        info["so_sha16"] = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
        side = json.load(open(so + ".buildinfo"))
        info["built_on_box"] = bool(side.get("host") == socket.gethostname() and side.get("gpu_visible_at_build"))
        info["built"] = {k: side.get(k) for k in ("host", "time", "gpu_visible_at_build", "hipcc")}
        if side.get("so_sha16") not in (None, info["so_sha16"]):
            info["built_on_box"] = None
            info["note"] = "sidecar does not describe this library"
    except Exception:
        pass
    return info


def cpu_baseline(task, sample_envs, sample_steps, threads=None):
    """Time the CPU oracle (OpenMP over envs/robots) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_desc
    from oracle_engine import OracleEngine, usable_cpus
    from mqe.engine import abi
    d, keep, _ = make_desc(task, sample_envs)
    e = OracleEngine(d, keep)
    # the oracle's OpenMP loops (over envs / robots) scale linearly up to the CPUs the container may actually use -- the GPU box shows 256
    # hardware threads but its cgroup grants 16 (cpu.max 1600000 100000: 11.6 k / 24.0 k / 34.4 k / 46.3 k env-steps/s on 4 / 8 / 12 / 16
    # threads, 37 k on 32, 2.6 k on 256) -- so the baseline runs on exactly that many threads, reported as `cores`
    if threads is None:
        threads = int(os.environ.get("MQE_CPU_THREADS", usable_cpus()))
    e.lib.mqo_set_num_threads(int(threads))
    e.reset_all()
    g = torch.Generator().manual_seed(1234)
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    acts = [torch.rand(sample_envs, Aw, 3, generator=g) * 2 - 1 for _ in range(sample_steps + 2)]
    e.step(acts[0]); e.step(acts[1])
    t0 = time.perf_counter()
    for t in range(sample_steps):
        e.step(acts[2 + t])
    dt = time.perf_counter() - t0
    e.lib.mqo_num_threads.restype = int
    return d.num_agents * sample_envs * sample_steps / dt, dt, int(e.lib.mqo_num_threads())


def time_variant(task, N, dev, steps, warmup, env_vars):
    """ms per step of a second engine instance created under `env_vars` (the switches are read at creation), same inputs"""
    from mqe.envs.utils import make_mqe_env, custom_cfg
    old = {k: os.environ.get(k) for k in env_vars}
    os.environ.update(env_vars)
    from mqe.envs.go1.go1 import Go1
    shard0, Go1.shard = Go1.shard, None       # a batch of its own, not a shard of the headline's (main() keys the scene by GLOBAL env ids: Go1.shard)
    try:
        margs = make_args(task, N, 0, dev)
        with contextlib.redirect_stdout(sys.stderr):
            env, _ = make_mqe_env(task, margs, custom_cfg(margs))
    finally:
        Go1.shard = shard0
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    A = env.num_agents
    gen = torch.Generator(device=dev).manual_seed(1234)
    env.reset()
    acts = [torch.rand(N, A, 3, device=dev, generator=gen) * 2 - 1 for _ in range(warmup + steps)]
    for t in range(warmup):
        env.step(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        env.step(acts[warmup + t])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    env.close()
    return 1e3 * el / steps, A


def event_pair_overhead_ms(n=32):
    """what a HIP event pair with NOTHING between its two records reports on the current stream: the share of a bracketed kernel's event time
    that is the bracket's own (subtracted from the per-kernel averages below; both figures are on the line)"""
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in pairs:
        a.record(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    return ts[len(ts) // 2]


def profile_pass(eng, step_fn, brackets=24, every=4):
    """HIP events around each kernel class on the launch stream, OUTSIDE every timed region (VERDICT r5 "Next" 7: the brackets perturb what
    they measure -- an event pair costs ~4 us of GPU timeline -- so no timed step carries one): `brackets` steps, one per period of `every`,
    of a run of brackets x every further steps of the same loop.  Returns (ms per class, launches per class, bracketed steps)."""
    eng.profile_enable(every)
    for _ in range(brackets * every):
        step_fn()
    ms, _ = eng.profile_read(12)
    eng.profile_enable(False)
    return ms[:6], ms[6:12], brackets


SUBSTEPS_BOUND = "valu-issue / latency (not hbm)"
SUBSTEPS_NOTE = ("achieved / peak / frac are the PRESCRIBED form -- the whole step's algorithmic HBM bytes over this kernel's average launch time against the HBM peak -- "
                 "but HBM is not what binds it: one env per wavefront, state LDS-resident for the 4 substeps; the limiters are the wavefront's chain of dependent "
                 "waits (~75 us for a wavefront alone on its CU) and the vector ALU's issue slots (the larger part of them taken with all envs resident at "
                 "4 waves/SIMD: valu_issue; round 6 took 12 % of the instructions out and the launch got 12 % shorter); DESIGN.md 3.1")

# the other BASELINE.json configs that fit one GPU (+ the headline with the URDF's exact thigh / calf boxes): timed on the driver's
# own line (`configs`), same protocol as the headline -- fresh U(-1,1) actions, inputs resident, device-synchronised
EXTRA_CONFIGS = [
    ("go1sheep-hard", 2048, {}, "BASELINE config 3"),
    ("go1seesaw", 4096, {}, "BASELINE config 4"),
    ("go1football-defender", 4096, {}, "BASELINE config 5's per-GPU shard (32768 envs over 8 GPUs)"),
    ("go1gate", 4096, {"MQE_COLLISION_MODEL": "exact"}, "BASELINE config 2 with the URDF's thigh / calf boxes (60 feature points per robot)"),
    ("go1gate", 4096, {"MQE_COLLISION_MODEL": "exact", "MQE_CONTACT_REDUCTION": "1"},
     "the same with the optional manifold reduction (desc.edge_contacts bit 8): a robot's contacts beyond its eight slots reduced to the deepest instead of truncated -- "
     "contact_overflow_substeps 0, contact_reduced_substeps instead; off by default (DESIGN.md section 0, round 6, item 4)"),
    ("go1gate", 8192, {}, "BASELINE config 2 at twice the batch (two residency rounds of k_substeps: 16 envs per CU are what 9.9 kB of LDS per env allow)"),
    ("go1gate", 16384, {}, "BASELINE config 2 at four times the batch"),
]


def time_config(task, N, dev, env_vars, label, min_ms=100.0, warmup=10):
    """One more workload on the same line: >= min_ms of timed steps of a fresh env created under `env_vars`, with the physics kernel's
    HIP-event time (every 16th step bracketed) and the prescribed roofline form on it."""
    from mqe.envs.utils import make_mqe_env, custom_cfg
    old = {k: os.environ.get(k) for k in env_vars}
    os.environ.update(env_vars)
    from mqe.envs.go1.go1 import Go1
    shard0, Go1.shard = Go1.shard, None       # a batch of its own, not a shard of the headline's (main() keys the scene by GLOBAL env ids: Go1.shard)
    try:
        margs = make_args(task, N, 0, dev)
        with contextlib.redirect_stdout(sys.stderr):
            env, _ = make_mqe_env(task, margs, custom_cfg(margs))
    finally:
        Go1.shard = shard0
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    A = env.num_agents
    A_phys = env.env.num_agents
    P = env.env.num_npcs
    eng = env.env.engine
    gen = torch.Generator(device=dev).manual_seed(1234)
    env.reset()
    pool = [torch.rand(N, A, 3, device=dev, generator=gen) * 2 - 1 for _ in range(64)]
    for t in range(warmup):
        env.step(pool[t % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(8):
        env.step(pool[(warmup + t) % 64])
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 8
    steps = int(min(5000, max(32, -(-min_ms * 1e-3 // est))))
    # (>= 100 ms timed: one step in `every` carries the brackets -- >= 16 of them, spread over the whole rollout, whose contact counts drift --
    # at a cost of ~16 us per bracketed step, < 0.5 % of the region)
    every = max(3, steps // 20)
    eng.profile_enable(every)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        env.step(pool[t % 64])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, _ = eng.profile_read(12)
    eng.profile_enable(False)
    kms, cnt = ms[:6], ms[6:12]
    overflow = int(env.env.contact_overflow.sum().item())
    reduced = int(env.env.engine.tensor(abi_mod().T_CONTACT_REDUCED).sum().item())
    ovh = event_pair_overhead_ms()
    kms = [max(kms[i] - ovh * cnt[i], 0.0) for i in range(6)]
    sub_ms = kms[3] / max(cnt[3], 1)
    R = N * A_phys
    step_bytes = 10128.0 * R + 2 * 104.0 * N * P
    ach = step_bytes / (sub_ms * 1e-3) / 1e9 if sub_ms > 0 else 0.0
    value = A * N * steps / el
    row = {"workload": f"{task}, {A} agents" + (f" (+ {A_phys - A} scripted)" if A_phys != A else "") + (f" + {P} NPC" if P else "") + f", num_envs={N}",
           "what": label, "switches": env_vars or None,
           "value": round(value, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps, "warmup": warmup + 8,
           "physical_robot_steps_per_s": round(value * A_phys / A, 1),
           "roofline": {"kernel": "k_substeps", "bound": SUBSTEPS_BOUND, "achieved": round(ach, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 6), "avg_launch_ms": round(sub_ms, 4), "launches": int(cnt[3]), "traffic": None,
                        "bytes_basis": "SURVEY 8(d): 10128 B per robot-step + 208 B per free NPC body and env step, charged to the dominant kernel"},
           "kernel_avg_launch_ms": {("k_gemm_h2", "k_policy_tail", "k_compute_torques_mfma", "k_substeps", "k_post_physics", "k_pre_policy")[i]: round(kms[i] / max(cnt[i], 1), 4)
                                    for i in range(6) if cnt[i] > 0},
           "contact_overflow_substeps": overflow, "contact_reduced_substeps": reduced,
           "hip_event_sampling": "one step in %d bracketed; event-pair overhead %.4f ms subtracted per launch" % (every, ovh)}
    env.close()
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--task", type=str, default="go1gate")
    ap.add_argument("--num_envs", type=int, default=4096, help="envs PER GPU (weak scaling)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_strict_f32", action="store_true", help="skip the exact-f32 companion run")
    ap.add_argument("--no_configs", action="store_true", help="skip the other single-GPU BASELINE configs (`configs` on the line)")
    ap.add_argument("--cpu_sample_envs", type=int, default=0, help="CPU baseline batch (0 = the GPU run's num_envs, 64 steps)")
    ap.add_argument("--cpu_sample_steps", type=int, default=64)
    ap.add_argument("--gather", choices=["between", "after", "tail"], default=os.environ.get("MQE_BENCH_GATHER", "tail"),
                    help="N > 1: where the all-gather of a step's returned batch is issued (default: tail).  between: inside the NEXT step, after its policy "
                         "kernels and before its physics kernel (overlaps k_substeps; DESIGN.md 8).  after: right behind the step's own k_post_physics, "
                         "and the next step's first kernel waits for it (no RCCL block is ever resident beside the two machine-filling kernels; "
                         "costs the gather's latency once per step).  tail: issued inside the next step after layer 0 of the policy, so that the RCCL kernel runs beside "
                         "k_policy_tail (one 4-wave workgroup per CU: room to spare), and the physics kernel waits for it (costs what the gather takes beyond the tail's 26 us)")
    ap.add_argument("--no_gather", action="store_true", help="N > 1: per-GPU learners -- every rank keeps its own batch, no collective at all (SURVEY 8e)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # experiment switch (1-GPU boxes): run the sharded code path -- RCCL group of ONE rank, the all-gather and its schedule -- to see
    # what each schedule costs on the compute stream; the JSON line then carries "collective" although n_gpus is 1
    solo_group = world == 1 and bool(os.environ.get("MQE_BENCH_WORLD1_GATHER"))
    if solo_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or solo_group:
        import torch.distributed as dist
        if os.environ.get("MQE_BENCH_SELFTEST_GLOO"):      # logic check of the sharded path on a 1-GPU box: all ranks on cuda:0, gloo
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            opts = None
            try:      # RCCL kernels on a high-priority HIP stream: they take their few CUs as soon as they are runnable
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                pass
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    from mqe.envs.utils import make_mqe_env, custom_cfg, ENV_DICT
    from mqe.envs.go1.go1 import Go1
    N = args.num_envs
    Go1.shard = (N * world, N * rank)           # global env ids -> identical scene regardless of the GPU count
    margs = make_args(args.task, N, 0, dev)
    with contextlib.redirect_stdout(sys.stderr):     # "Setting seed: 0" (as upstream, helpers.py:82) must not precede the ONE JSON line on stdout
        env, cfg = make_mqe_env(args.task, margs, custom_cfg(margs))
    A = env.num_agents
    eng = env.env.engine
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    obs = env.reset()
    gather = None
    pending = [None, None]
    sent = [None, None]                          # the snapshot each in-flight gather reads (kept alive until it is waited for)
    if world > 1 or solo_group:
        from mqe.engine import abi
        gather = [torch.empty(world * eng.tensor(abi.T_WRAPPER_PACKED).numel(), device=dev) for _ in range(2)]   # double buffer, [world][L]

    # synthetic inputs: one fresh U(-1,1) action tensor per step, generated before the clock starts (the contract times the
    # hot path with its inputs already resident in HBM; 98 kB per step)
    actions = [torch.rand(N, A, 3, device=dev, generator=gen) * 2 - 1 for _ in range(args.warmup + args.steps)]
    step_no = [0]

    # The one collective of the path: all-gather of the returned batch, asynchronous and double-buffered.  What is gathered
    # is the snapshot env.step() makes anyway (MQE_T_WRAPPER_PACKED: obs | reward | done, one copy), so the sharded step has
    # no kernels of its own on the compute stream.  The gather of step t is ISSUED inside step t+1, between the policy
    # kernels and the physics kernel (Go1.between_policy_and_physics -> mqe_step_begin / mqe_step_end): the comm stream then
    # waits for the policy of step t+1 and the RCCL kernel shares the GPU with k_substeps (thousands of independent waves:
    # a few displaced CUs cost their share of 0.2 ms) instead of with the layer-0 GEMM (one workgroup per CU: one displaced
    # workgroup = a second round).
    ready = [None]                # (buffer index, snapshot) of the previous step, not gathered yet
    n_gathers = [0]

    def issue_gather():
        if ready[0] is not None:
            b, snap = ready[0]
            if pending[b] is not None:
                pending[b].wait()                     # the gather that filled this buffer two steps ago (stream-side wait)
            sent[b] = snap
            pending[b] = dist.all_gather_into_tensor(gather[b], snap, async_op=True)
            ready[0] = None
            n_gathers[0] += 1

    sharded = world > 1 or solo_group
    sched = ["none" if (args.no_gather or not sharded) else args.gather]       # the schedule in effect (MQE_BENCH_SWEEP_GATHER re-times the others)

    def wait_gathers():               # stream-side: what the compute stream launches next waits for the collectives in flight
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def set_schedule(name):
        sched[0] = name
        env.env.before_policy_tail = issue_gather if name == "tail" else None
        env.env.between_policy_and_physics = issue_gather if name == "between" else (wait_gathers if name == "tail" else None)
    set_schedule(sched[0])

    def one_step():
        t = step_no[0]
        a = actions[t % len(actions)]
        step_no[0] += 1
        use_gather = sched[0] != "none"
        if use_gather and sched[0] == "after":
            # schedule "after": the gather of the previous batch was issued behind that step's last kernel; this step's first
            # kernel waits for it (stream-side), so the RCCL kernel never shares the GPU with the GEMM or the physics kernel
            for b in range(2):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
        o, r, d, info = env.step(a)
        if use_gather:
            ready[0] = (t & 1, env.returned_batch)
            if sched[0] == "after":
                issue_gather()
        return o

    def drain():
        if sched[0] != "none":
            issue_gather()                            # the last step's batch
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        one_step()
    drain()
    # (no HIP-event brackets inside any timed region: profile_pass() below runs after the clocks have stopped)
    prof_on = not os.environ.get("MQE_BENCH_NOPROF")

    def timed(n):
        """n steps bracketed by barrier + synchronize on both sides; seconds, MAX over ranks"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            one_step()
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el
    elapsed = timed(args.steps)
    use_gather = sched[0] != "none"
    if use_gather:   # every step's batch was gathered, and the last one arrived whole: this rank's slice is its own snapshot
        b = (args.warmup + args.steps - 1) & 1
        assert n_gathers[0] == args.warmup + args.steps, (n_gathers[0], args.warmup + args.steps)
        mine = gather[b].view(world, -1)[rank]
        assert torch.equal(mine.view(torch.int32), env.returned_batch.view(torch.int32)), "all-gather: own slice differs from the returned batch"
        assert torch.equal(mine[obs.numel() + N * A:].view(torch.uint8)[:N].view(torch.bool), env.env.reset_buf), "all-gather: done flags"
    gathers_contract = n_gathers[0]              # collectives of the warm-up + the K timed steps (the long run below adds its own)
    overflow_substeps = int(env.env.contact_overflow.sum().item())      # truncated contact lists over the whole run (after the clock stopped)
    reduced_substeps = int(eng.tensor(abi_mod().T_CONTACT_REDUCED).sum().item())   # env-substeps with a robot's contacts reduced to its deepest eight
    value = A * N * world * args.steps / elapsed
    # A timed region shorter than 50 ms (the driver's --steps 20 is 4.6 ms) is below the resolution of everything that watches the
    # box from outside (rocm-smi samples, the driver's own clock): the same loop is continued until >= 100 ms of GPU time have been
    # timed and reported beside the contract's number as value_long (value itself stays the K steps the caller asked for).
    long_run = None
    if elapsed < 0.05 and not os.environ.get("MQE_BENCH_NO_LONG"):
        n_long = int(min(20000, max(args.steps, -(-0.1 // (elapsed / args.steps)))))
        el_long = timed(n_long)
        long_run = {"steps": n_long, "ms_per_step": round(1e3 * el_long / n_long, 4), "value": round(A * N * world * n_long / el_long, 1),
                    "note": "the timed loop continued (same engine, same schedule, fresh actions cycled) until >= 100 ms were timed"}
    # per-kernel-class HIP-event times: 24 bracketed steps of the same loop, after both clocks have stopped
    ms = [0.0] * 12
    sampled = 1
    pair_ms = 0.0
    if prof_on:
        kms_, cnt_, sampled = profile_pass(eng, one_step)
        drain()
        pair_ms = event_pair_overhead_ms()
        ms = list(kms_) + list(cnt_)
    # MQE_BENCH_SWEEP_GATHER=1 (N > 1): one run decides the default schedule -- every all-gather schedule and the no-collective mode
    # re-timed back to back on the same engine, ms per step each (the headline above is the schedule named in `collective`)
    sweep = None
    if sharded and not args.no_gather and os.environ.get("MQE_BENCH_SWEEP_GATHER"):
        sweep = {}
        n_sw = max(args.steps, 200)
        for name in ("none", "tail", "between", "after"):
            set_schedule(name)
            timed(20)
            sweep[name] = round(1e3 * timed(n_sw) / n_sw, 4)
        set_schedule("none" if args.no_gather else args.gather)
        base_ms = sweep["none"]
        sweep = {"steps": n_sw, "ms_per_step": sweep, "overhead_vs_no_collective": {k: round(v / base_ms - 1.0, 4) for k, v in sweep.items() if k != "none"}}

    if rank == 0:
        kms_raw = ms[:6]
        cnt = ms[6:12]
        kms = [max(kms_raw[i] - pair_ms * cnt[i], 0.0) for i in range(6)]      # the bracket's own share out (event_pair_overhead_ms)
        tot = sum(kms) or 1.0
        dom = max(range(6), key=lambda i: kms[i])
        R = N * A
        P = env.env.num_npcs
        split = os.environ.get("MQE_GEMM_SPLIT", "1") != "0"
        d = eng.desc
        h_a, h_b = d.adaptation.dims[1], d.body.dims[1]

        def avg_ms(i):
            return kms[i] / max(cnt[i], 1)

        def launches_per_step(i):
            return max(1.0, cnt[i] / sampled)
        # ---- policy layer 0: the one MFMA-bound kernel
        l0_ms = max(avg_ms(0), 1e-9)
        flops32 = 2.0 * R * 2100 * (h_a + h_b)                        # algorithmic: unpadded K = 30 x 70, f32 products
        if split:   # every f32 product = three f16 x f16 terms on the matrix cores, K = 30 compact frames of MQE_H2_FRAME = 48 columns (mqe_common.hpp)
            l0 = {"kernel": "k_gemm_h2 (fused layer 0 of adaptation+body MLP over the history ring; 2-plane split-f16 operands, 3 MFMA terms per product, f32-class accuracy)",
                  "bound": "mfma", "achieved": round(3 * 2.0 * R * K_SPLIT * (h_a + h_b) / (l0_ms * 1e-3) / 1e12, 2), "peak": PEAK_F16_MFMA_TFLOPS,
                  "unit": "TFLOP/s", "f32_equivalent_TFLOPs": round(flops32 / (l0_ms * 1e-3) / 1e12, 2), "executed_K": K_SPLIT,
                  "flops_basis": "executed f16 MFMA flops: 3 terms x 2 R (256 + 512) x K, K = 30 frames x 48 compact columns (12 constant and 12 repeated columns of the 70 are folded into the weights)",
                  "note": "the peak assumes 2.4 GHz; measured with in-kernel clocks (profiles/r05_gemm_h2_bound.txt, not in this run): the shader clock is ~1.56 GHz during this kernel's K loop "
                          "(2.3 GHz when the same loop multiplies zeros: same tick count) -- the matrix pipe is busy 73 % of the loop's shader-clock ticks, 62 % of the kernel's"}
        else:
            l0 = {"kernel": "k_gemm_f32 (fused layer 0, exact f32 MFMA)", "bound": "mfma", "achieved": round(flops32 / (l0_ms * 1e-3) / 1e12, 3),
                  "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s"}
        l0.update({"frac": round(l0["achieved"] / l0["peak"], 4), "traffic": None, "avg_launch_ms": round(l0_ms, 4), "launches": int(cnt[0])})
        # ---- headline roofline (prescribed form): the WHOLE step's algorithmic bytes (10128 B per agent-step, + 104 B
        # read/write per free NPC body and env step) over the dominant kernel's average launch time
        step_bytes = 10128.0 * R + 2 * 104.0 * N * P
        if dom == 0:
            roof = dict(l0)
        else:
            byts = step_bytes / launches_per_step(dom)
            ach = byts / (avg_ms(dom) * 1e-3) / 1e9
            roof = {"kernel": PROF_NAMES[dom], "bound": SUBSTEPS_BOUND if dom == 3 else "hbm", "achieved": round(ach, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms(dom), 4), "launches": int(cnt[dom]),
                    "bytes_basis": "SURVEY 8(d): the whole step's 10128 B/agent-step charged to the dominant kernel; per-kernel shares in roofline_per_kernel",
                    "note": SUBSTEPS_NOTE}
        # HBM traffic of the dominant kernel: NOT measured in this run -- copied from the committed rocprofv3 PMC passes (separate
        # runs of this command, profiles/*pmc_summary.json) and labelled as such
        try:
            import glob
            pfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")))[-1]
            pmc = json.load(open(pfile))
            if N == 4096 and args.task == "go1gate":
                kname = ("k_gemm_h2" if split else "k_gemm_f32") if dom == 0 else PROF_KERNEL[dom]
                e = pmc.get(kname, {})
                roof["traffic"] = (e.get("fetch_bytes_x2", 0) + e.get("write_bytes_raw", 0)) if dom == 0 else e.get("hbm_bytes_raw")
                roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, bytes per launch, copied from " + os.path.relpath(pfile, ROOT)
                roof["pmc_from_profile"] = {"file": os.path.relpath(pfile, ROOT), "measured_in_this_run": False,
                                            **{k: e[k] for k in ("frac_wave_time_issuing", "frac_wave_time_issue_stalled", "frac_wave_time_waiting_on_waitcnt_or_barrier") if k in e}}
                if dom != 0 and e.get("SQ_ACTIVE_INST_VALU_mean") and e.get("GRBM_GUI_ACTIVE_mean"):
                    # the limit this kernel actually runs against: the vector ALU's issue slots.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the
                    # wavefronts of a launch; 1024 SIMDs issue one VALU instruction per quad-cycle each; GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles
                    simd_quads = e["GRBM_GUI_ACTIVE_mean"] / 8.0 / 4.0 * 1024.0
                    per_sub = None
                    try:      # VALU instructions per wavefront and substep: the last line of the newest per-phase counter table
                        pc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*phase_counters_go1gate_tgs.txt")))[-1]
                        hdr = open(pc).readline().split()
                        row = [l for l in open(pc) if l.startswith("whole substep")][-1].split()
                        per_sub = float(row[2 + hdr.index("INSTS_VALU") - 1])
                    except Exception:
                        pass
                    roof["valu_issue"] = {"bound": "valu issue slots (1 per SIMD and quad-cycle)", "frac_of_launch": round(e["SQ_ACTIVE_INST_VALU_mean"] / simd_quads, 3),
                                          "valu_instructions_per_wavefront_launch": round(e["SQ_INSTS_VALU_mean"] / max(e.get("SQ_WAVES_mean", 1.0), 1.0), 1) if e.get("SQ_INSTS_VALU_mean") else None,
                                          "valu_instructions_per_wavefront_substep": per_sub, "measured_in_this_run": False,
                                          "note": "SQ_ACTIVE_INST_VALU over the launch's SIMD quad-cycles (profiles/*pmc_summary.json); per substep from profiles/*phase_counters_go1gate_tgs.txt; "
                                                  "the launch-wide figure includes the state load, the epilogue and the wait for the last wavefront; DESIGN.md 3.1"}
        except Exception:
            pass
        # ---- per-kernel table: every kernel class with ITS OWN share of the algorithmic bytes (the items of SURVEY 8(d) assigned
        # to the kernel that moves them) resp. its flops, against its own average launch time measured in this run
        own_bytes = {   # per agent-step: (read, written)
            5: (12 + 96, 280),                    # k_pre_policy: command, last two actions -> history frame
            0: (8400, 0),                         # layer 0: the 30 x 70 float history
            1: (0, 96),                           # policy tail: last actions
            3: (52 + 96 + 192, 52 + 96 + 192 + 204),   # k_substeps: root, dof, actuator history -> the same + net contact forces
            4: (4, 4 + 284 + 64 + 4),             # k_post_physics: gait index -> gait, obs bag, wrapper obs, reward
        }
        if cnt[4] <= 0:      # the post-physics step ran as k_substeps' epilogue: its bytes are that launch's
            own_bytes[3] = (own_bytes[3][0] + own_bytes[4][0], own_bytes[3][1] + own_bytes[4][1])
        tail_flops = 2.0 * R * (sum(d.adaptation.dims[l] * d.adaptation.dims[l + 1] for l in range(1, d.adaptation.n_layers)) +
                                sum(d.body.dims[l] * d.body.dims[l + 1] for l in range(1, d.body.n_layers)) + 2 * h_b)
        act_flops = 2.0 * R * 12 * 4 * 1248
        table = []
        for i in (5, 0, 1, 3, 4):
            if cnt[i] <= 0:
                continue
            rd, wr = own_bytes[i]
            npc = 2 * 104.0 * N * P if i == 3 else 0.0
            byts = ((rd + wr) * R + npc) / launches_per_step(i)
            t = avg_ms(i) * 1e-3
            row = {"kernel": PROF_NAMES[i], "avg_launch_ms": round(avg_ms(i), 4), "launches_per_step": round(launches_per_step(i), 2),
                   "algorithmic_bytes_per_launch": int(byts), "hbm_GBps": round(byts / t / 1e9, 2), "hbm_frac": round(byts / t / 1e9 / PEAK_HBM_GBS, 5)}
            if i == 0:
                row.update({"bound": "mfma", "TFLOPs": l0["achieved"], "mfma_frac": l0["frac"], "peak_TFLOPs": l0["peak"]})
            elif i == 1:
                row.update({"bound": "latency (L2 weight stream, 6 dependent stages)", "f32_equivalent_TFLOPs": round(tail_flops / launches_per_step(i) / t / 1e12, 2)})
            elif i == 3:
                row.update({"bound": SUBSTEPS_BOUND, "actuator_mlp_f32_mfma_TFLOPs": round(act_flops / t / 1e12, 2)})
            else:
                row.update({"bound": "latency / hbm"})
            table.append(row)
        out = {
            "metric": f"env-steps/sec (agents x envs x steps/s), {args.task} {N} envs x {A} agents per GPU",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (policy GEMMs and the actuator network's 32 x 32 layer: 2-plane split-f16, 22-bit)" if split else "f32", "data": "synthetic (U(-1,1) actions seed 1234; synthetic body MLP: body_latest.jit missing upstream)",
            "config": {"workload": f"{args.task}, {A} agents, num_envs={N} per GPU ({N * world} total), 4 substeps x 5 ms per step",
                       "parallelism": (f"env-sharded x{world}, " + ("no collective (per-GPU learners)" if args.no_gather else
                                       f"all-gather of the returned batch issued {dict(between='between policy and physics of the next step', after='after the step, next step waits', tail='after layer 0 of the next step (beside the policy tail), physics waits')[args.gather]}"))
                                      if world > 1 else "single GPU"},
            "collective": None if (world == 1 and not solo_group) else ("none" if args.no_gather else {"op": "all_gather_into_tensor", "schedule": args.gather, "bytes_per_rank": int(4 * env.returned_batch.numel()),
                                                                                  "gathers": gathers_contract, "gathers_incl_long_run_and_sweep": n_gathers[0]}),
            "target_env_steps_per_s": 1.0e6,
            "contact_solver": {0: "pgs: velocity-level projected Gauss-Seidel, erp %.2f" % d.erp, 1: "tgs: temporal Gauss-Seidel, %d sub-steps of dt / %d (sim.physx.solver_type = 1)" % (d.solver_iterations, d.solver_iterations)}[int(d.solver_type)],
            "value_long": long_run["value"] if long_run else None, "ms_per_step_long": long_run["ms_per_step"] if long_run else None, "long_run": long_run,
            "gather_schedule_sweep": sweep,
            "binary": binary_info(),
            "physical_robot_steps_per_s": round(value * env.env.num_agents / A, 1),
            "roofline": roof,
            "roofline_policy_layer0": l0,
            "roofline_per_kernel": table,
            "hbm_step_algorithmic_GBps": round(step_bytes * args.steps / elapsed / 1e9, 3),
            "hbm_step_algorithmic_frac": round(step_bytes * args.steps / elapsed / 1e9 / PEAK_HBM_GBS, 5),
            "kernel_time_share": {PROF_NAMES[i]: round(kms[i] / tot, 4) for i in range(6)},
            "gpu_busy_ms_per_step": round(tot / sampled, 4),
            "gpu_busy_ms_per_step_raw_event_times": round(sum(kms_raw) / sampled, 4),
            # the kernel classes' GPU time per step cannot exceed the un-bracketed step (which also holds the gaps between the launches)
            "gpu_busy_le_ms_per_step": bool(tot / sampled <= (long_run["ms_per_step"] if long_run else 1e3 * elapsed / args.steps) * 1.001) if prof_on else None,
            "contact_overflow_substeps": overflow_substeps,      # env-substeps in which a touching pair found no slot (of envs x 4 x steps)
            "contact_reduced_substeps": reduced_substeps,        # env-substeps in which a robot's one-sided contacts were reduced to its deepest eight
            "hip_event_sampling": (f"{sampled} steps bracketed (one per period of 4) AFTER the timed regions; an empty event pair reads {pair_ms:.4f} ms on this stream, "
                                   "subtracted per launch (raw sums beside)") if prof_on else "off",
            "dtype_note": "state and physics: f32.  Policy layer 0 + tail and (round 5) layer 2 of the actuator network: f32 operands carried as two f16 planes "
                          "(22 significand bits, 3 MFMA terms), held to the same 5e-5 bound as exact f32; strict_f32 = the same run on the exact-f32 kernels",
        }
        if world == 1 and not args.no_strict_f32 and not os.environ.get("MQE_BENCH_NOPROF"):
            # the dtype claim's companion: identical workload with every policy GEMM on the exact-f32 kernels
            sms, _ = time_variant(args.task, N, dev, min(args.steps, 100), min(args.warmup, 10), {"MQE_GEMM_SPLIT": "0", "MQE_NO_FUSED_TAIL": "1", "MQE_ACT_F32": "1"})
            out["strict_f32"] = {"value": round(A * N / (sms * 1e-3), 1), "unit": "env-steps/s", "ms_per_step": round(sms, 4), "steps": min(args.steps, 100),
                                 "switches": "MQE_GEMM_SPLIT=0 MQE_NO_FUSED_TAIL=1 (k_gemm_f32 for every policy layer) MQE_ACT_F32=1 (actuator network: the f32 MFMA chain)"}
        if world == 1 and not args.no_strict_f32 and not os.environ.get("MQE_BENCH_NOPROF") and not os.environ.get("MQE_SOLVER"):
            # what the choice of contact solver costs: the same workload on the velocity-level sweeps of rounds 1-3 (solver_type 0)
            pms, _ = time_variant(args.task, N, dev, min(args.steps, 100), min(args.warmup, 10), {"MQE_SOLVER": "pgs"})
            out["solver_pgs"] = {"value": round(A * N / (pms * 1e-3), 1), "unit": "env-steps/s", "ms_per_step": round(pms, 4), "steps": min(args.steps, 100),
                                 "switches": "MQE_SOLVER=pgs (desc.solver_type = 0: velocity-level projected Gauss-Seidel with the erp bias)"}
        lean = args.no_strict_f32 and args.no_cpu_baseline          # the A/B and profiling scripts under tools/: the headline alone
        if world == 1 and not args.no_configs and not lean and not os.environ.get("MQE_BENCH_NOPROF") and args.task == "go1gate" and N == 4096:
            # absolute env-steps/s and the roofline fraction of every config that fits one GPU, on the driver's line
            out["configs"] = [time_config(t, n, dev, ev, what) for (t, n, ev, what) in EXTRA_CONFIGS]
        if not args.no_cpu_baseline and world == 1:
            torch.set_num_threads(os.cpu_count() or 1)
            # the CPU restatement at the headline size (few steps) and at the reference's own CPU-runnable size (N = 4)
            big_n, big_steps = (args.cpu_sample_envs, args.cpu_sample_steps) if args.cpu_sample_envs else (N, 64)
            v, secs, nthr = cpu_baseline(args.task, big_n, big_steps)
            out["cpu_baseline"] = {"value": round(v, 1), "unit": "env-steps/s", "cores": nthr, "host_threads_available": os.cpu_count(), "host_cpus_usable": nthr if not os.environ.get("MQE_CPU_THREADS") else None,
                                   "kind": "port", "sample": f"{args.task} {big_n} envs x {big_steps} steps, build's CPU restatement (oracle/, OpenMP over envs), {secs:.1f} s"}
            v4, secs4, nthr4 = cpu_baseline(args.task, 4, 2000, threads=4)
            out["cpu_baseline_n4"] = {"value": round(v4, 1), "unit": "env-steps/s", "cores": nthr4, "kind": "port",
                                      "sample": f"{args.task} 4 envs x 2000 steps (BASELINE config 1 size), {secs4:.1f} s"}
        print(json.dumps(out))
    env.close()
    if world > 1 or solo_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
