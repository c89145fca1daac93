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
`roofline`: dominant kernel (largest share of GPU time, HIP events on the launch stream over the timed region).
`cpu_baseline`: the build's CPU restatement (oracle/, kind "port") timed on this box's host cores on a bounded sample.
NOTE: body_latest.jit is missing from the reference snapshot, so the locomotion-policy body is the deterministic
synthetic 2102-512-256-128-12 ELU MLP (mqe/utils/policy_weights.py); the adaptation module and actuator net are real.
"""
import argparse
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


def make_args(task, num_envs, seed, device):
    from mqe.utils.helpers import finish_args
    a = types.SimpleNamespace(task=task, num_envs=num_envs, seed=seed, headless=True, record_video=False,
                              sim_device=device, pipeline="gpu", subscenes=0, num_threads=0)
    return finish_args(a)


def cpu_baseline(task, sample_envs, sample_steps):
    """Time the CPU oracle (OpenMP over envs/robots) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_desc
    from oracle_engine import OracleEngine
    from mqe.engine import abi
    d, keep, _ = make_desc(task, sample_envs)
    e = OracleEngine(d, keep)
    # the oracle's OpenMP loops (over envs / robots) peak around 32 threads on the box's 256 hardware threads and
    # collapse when oversubscribed, so the baseline is timed at its best setting, which is reported as `cores`
    e.lib.mqo_set_num_threads(int(os.environ.get("MQE_CPU_THREADS", min(32, os.cpu_count() or 1))))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--task", type=str, default="go1gate")
    ap.add_argument("--num_envs", type=int, default=4096, help="envs PER GPU (weak scaling)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_sample_envs", type=int, default=512)
    ap.add_argument("--cpu_sample_steps", type=int, default=100)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
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
    env, cfg = make_mqe_env(args.task, margs, custom_cfg(margs))
    A = env.num_agents
    eng = env.env.engine
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    obs = env.reset()
    gather = None
    pending = [None, None]
    sent = [None, None]                          # the snapshot each in-flight gather reads (kept alive until it is waited for)
    if world > 1:
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

    if world > 1:
        env.env.between_policy_and_physics = issue_gather

    def one_step():
        t = step_no[0]
        a = actions[t]
        step_no[0] += 1
        o, r, d, info = env.step(a)
        if world > 1:
            ready[0] = (t & 1, env.returned_batch)
        return o

    def drain():
        if world > 1:
            issue_gather()                            # the last step's batch
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        one_step()
    drain()
    # HIP events around each kernel class on the launch stream, on every PROF_EVERY-th step of the timed region (an event
    # pair costs ~4 us of GPU timeline; bracketing all 5 classes of every step would inflate ms_per_step by 7 %)
    prof_every = 0 if os.environ.get("MQE_BENCH_NOPROF") else max(1, int(os.environ.get("MQE_BENCH_PROF_EVERY", "16")))
    eng.profile_enable(prof_every)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:   # every step's batch was gathered, and the last one arrived whole: this rank's slice is its own snapshot
        b = (args.warmup + args.steps - 1) & 1
        assert n_gathers[0] == args.warmup + args.steps, (n_gathers[0], args.warmup + args.steps)
        mine = gather[b].view(world, -1)[rank]
        assert torch.equal(mine, env.returned_batch), "all-gather: own slice differs from the returned batch"
        assert torch.equal(mine[obs.numel() + N * A:] != 0, env.env.reset_buf), "all-gather: done flags"
    ms, _ = eng.profile_read(12)
    eng.profile_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = A * N * world * args.steps / elapsed

    if rank == 0:
        kms = ms[:6]
        cnt = ms[6:12]
        tot = sum(kms) or 1.0
        dom = max(range(6), key=lambda i: kms[i])
        R = N * A
        roof = None
        split = os.environ.get("MQE_GEMM_SPLIT", "1") != "0"
        d = eng.desc
        h_a, h_b = d.adaptation.dims[1], d.body.dims[1]
        l0_ms = max(kms[0] / max(cnt[0], 1), 1e-9)
        flops32 = 2.0 * R * 2100 * (h_a + h_b)                        # algorithmic: unpadded K = 30 x 70, f32 products
        if split:   # every f32 product = three f16 x f16 terms on the matrix cores, K padded to 2208 (a multiple of three 32-k tiles)
            l0 = {"kernel": "k_gemm_h2 (fused layer 0 of adaptation+body MLP over the history ring; 2-plane split-f16 operands, 3 MFMA terms per product, f32-class accuracy)",
                  "bound": "mfma", "achieved": round(3 * 2.0 * R * 2208 * (h_a + h_b) / (l0_ms * 1e-3) / 1e12, 2), "peak": PEAK_F16_MFMA_TFLOPS,
                  "unit": "TFLOP/s", "f32_equivalent_TFLOPs": round(flops32 / (l0_ms * 1e-3) / 1e12, 2)}
        else:
            l0 = {"kernel": "k_gemm_f32 (fused layer 0, exact f32 MFMA)", "bound": "mfma", "achieved": round(flops32 / (l0_ms * 1e-3) / 1e12, 3),
                  "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s"}
        l0.update({"frac": round(l0["achieved"] / l0["peak"], 4), "traffic": None, "avg_launch_ms": round(l0_ms, 4), "launches": int(cnt[0])})
        if dom == 0:
            roof = l0
        else:
            avg_ms = kms[dom] / max(cnt[dom], 1)
            sampled = max(1, -(-args.steps // max(prof_every, 1)))     # env steps whose launches were bracketed
            per_launch = max(1.0, cnt[dom] / sampled)                 # launches of this kernel class per env step
            byts = 10128.0 * R / per_launch                           # 10128 B per agent-step
            ach = byts / (avg_ms * 1e-3) / 1e9
            roof = {"kernel": PROF_NAMES[dom], "bound": "hbm", "achieved": round(ach, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": int(cnt[dom]),
                    "note": "not HBM-bound by construction: one env per wavefront, state LDS-resident for the 4 substeps; the limiter is "
                            "dependent-instruction latency at 2 waves/SIMD (see profiles/*pmc_summary.json: issue / wait fractions)"}
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (separate runs; profiles/*pmc_summary.json)
        try:
            import glob
            pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")))[-1]))
            if N == 4096 and args.task == "go1gate":
                kname = ("k_gemm_h2" if split else "k_gemm_f32") if dom == 0 else PROF_KERNEL[dom]
                e = pmc.get(kname, {})
                roof["traffic"] = (e.get("fetch_bytes_x2", 0) + e.get("write_bytes_raw", 0)) if dom == 0 else e.get("hbm_bytes_raw")
                for k in ("frac_wave_time_issuing", "frac_wave_time_issue_stalled", "frac_wave_time_waiting_on_waitcnt_or_barrier"):
                    if k in e:
                        roof[k] = e[k]
                roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch, see profiles/"
        except Exception:
            pass
        step_bytes = 10128.0 * R
        out = {
            "metric": f"env-steps/sec (agents x envs x steps/s), {args.task} {N} envs x {A} agents per GPU",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (U(-1,1) actions seed 1234; synthetic body MLP: body_latest.jit missing upstream)",
            "config": {"workload": f"{args.task}, {A} agents, num_envs={N} per GPU ({N * world} total), 4 substeps x 5 ms per step",
                       "parallelism": f"env-sharded x{world}, all-gather of the returned batch" if world > 1 else "single GPU"},
            "target_env_steps_per_s": 1.0e6,
            "physical_robot_steps_per_s": round(value * env.env.num_agents / A, 1),
            "roofline": roof,
            "roofline_policy_layer0": l0,
            "hbm_step_algorithmic_GBps": round(step_bytes * args.steps / elapsed / 1e9, 3),
            "kernel_time_share": {PROF_NAMES[i]: round(kms[i] / tot, 4) for i in range(6)},
            "gpu_busy_ms_per_step": round(tot / max(1, -(-args.steps // max(prof_every, 1))), 4),
            "hip_event_sampling": f"kernel classes of every {prof_every}-th timed step bracketed" if prof_every else "off",
        }
        if not args.no_cpu_baseline and world == 1:
            torch.set_num_threads(os.cpu_count() or 1)
            v, secs, nthr = cpu_baseline(args.task, args.cpu_sample_envs, args.cpu_sample_steps)
            out["cpu_baseline"] = {"value": round(v, 1), "unit": "env-steps/s", "cores": nthr,
                                   "kind": "port", "sample": f"{args.task} {args.cpu_sample_envs} envs x {args.cpu_sample_steps} steps, build's CPU restatement (oracle/), {secs:.1f} s"}
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
