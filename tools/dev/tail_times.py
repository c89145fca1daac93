#!/usr/bin/env python3
"""Where k_policy_tail's time goes: every workgroup stamps the 100 MHz wall clock at its entry, after each barrier and at its end
(MQE_TAIL_TIMES=1 at creation, mqe_debug_tail_times).  Usage: python tools/dev/tail_times.py [task] [num_envs] [steps]"""
import os
import sys
os.environ["MQE_TAIL_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-quadruped-environment_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import make_desc, hip_engine  # noqa: E402
from mqe.engine import abi  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "go1gate"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
d, k, _ = make_desc(task, N)
e = hip_engine(d, k)
e.reset_all()
Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
g = torch.Generator(device="cuda").manual_seed(1)
nb = (N * d.num_agents + 31) // 32
acc = []
gacc = []                                                          # k_gemm_h2's stamps (slots 10..15): wall x 3, shader clock x 3
for t in range(steps):
    e.step(torch.rand(N, Aw, 3, device="cuda", generator=g) * 2 - 1)
    if t >= steps - 20:
        out = np.zeros((nb, 16), np.int64)
        e._call("debug_tail_times", C.c_void_p(out.ctypes.data))
        acc.append(out[:, :10].astype(np.float64) * 0.01)          # us
        gacc.append(out[:, 10:16].astype(np.float64))
a = np.stack(acc)                                                  # [launch][block][stamp]
names = ["entry -> P1 rows requested, h0 split (stage 0)", "stage 1 (h1: 256 -> 128, waves 0-3)", "stage 2 MFMA (latent, wave 0)", "latent bias / store",
         "stage 3 (b0 = ELU(pre0 + latent w))", "stage 4 (b1: 512 -> 256, all waves)", "stage 5 (b2: 256 -> 128, waves 0-3)", "stage 6 MFMA (targets, wave 0)", "registers, stores"]
t0 = a[:, :, 0].min(axis=1, keepdims=True)
print(f"{task} {N} envs: k_policy_tail, mean over workgroups and the last {len(acc)} launches [us]")
for i in range(9):
    print(f"  {names[i]:52s} {np.mean(a[:, :, i + 1] - a[:, :, i]):6.2f}")
print(f"  {'workgroup entry .. exit':52s} {np.mean(a[:, :, 9] - a[:, :, 0]):6.2f}")
print(f"  {'first entry .. last exit of the launch':52s} {np.mean(a[:, :, 9].max(axis=1) - t0[:, 0]):6.2f}   (entries spread over {np.mean(a[:, :, 0].max(axis=1) - t0[:, 0]):.2f})")
gg = np.stack(gacc)
w = gg[:, :, 0:3] * 0.01
c = gg[:, :, 3:6]
loop_us, epi_us = np.mean(w[:, :, 1] - w[:, :, 0]), np.mean(w[:, :, 2] - w[:, :, 1])
loop_ck, epi_ck = np.mean(c[:, :, 1] - c[:, :, 0]), np.mean(c[:, :, 2] - c[:, :, 1])
print(f"k_gemm_h2 (same launches): K loop {loop_us:.2f} us = {loop_ck:.0f} shader-clock ticks ({loop_ck / loop_us * 1e-3:.3f} GHz), epilogue {epi_us:.2f} us = {epi_ck:.0f} ticks"
      f" ({epi_ck / epi_us * 1e-3:.3f} GHz); first entry .. last exit {np.mean(w[:, :, 2].max(axis=1) - w[:, :, 0].min(axis=1)):.2f} us;"
      f" gap to the tail's first entry {np.mean(a[:, :, 0].min(axis=1) - w[:, :, 2].max(axis=1)):.2f} us")
