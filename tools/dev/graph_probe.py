#!/usr/bin/env python3
"""Does a hipGraph shorten the step?  Timing probe only (the captured launches replay with FROZEN arguments -- ring slot, step counter --
so the rollout is not a valid one; the kernels and their order are the real ones): go1gate 4096 x 2, fused mqe_step
  (a) launched kernel by kernel, (b) one captured step replayed, (c) four captured steps per graph.
Usage: python tools/dev/graph_probe.py [task] [num_envs] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-quadruped-environment_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import make_desc, hip_engine  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "go1gate"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
d, keep, _ = make_desc(task, N)
e = hip_engine(d, keep)
e.reset_all()
from mqe.engine import abi  # noqa: E402
Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
g = torch.Generator(device="cuda").manual_seed(1)
acts = [torch.rand(N, Aw, 3, device="cuda", generator=g) * 2 - 1 for _ in range(16)]
for t in range(40):
    e.step(acts[t % 16])
torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


plain = timed(lambda i: e.step(acts[i % 16]), steps)
side = torch.cuda.Stream()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=side):
    e.step(acts[0])
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4, stream=side):
    for k in range(4):
        e.step(acts[k])
for _ in range(10):
    g1.replay()
one = timed(lambda i: g1.replay(), steps)
four = timed(lambda i: g4.replay(), steps // 4) / 4
plain2 = timed(lambda i: e.step(acts[i % 16]), steps)
print(f"{task} {N}: plain {plain:.4f} / {plain2:.4f} ms per step, graph of one step {one:.4f}, graph of four steps {four:.4f}")
