#!/usr/bin/env python3
"""Would two half batches on two streams fill the chip better than one whole batch?  Timing probe: the same task as
  (a) ONE handle of N envs stepped on one stream,
  (b) TWO handles of N / 2 envs, each stepped on a stream of its own (their launches interleave on the GPU: the MFMA- and LDS-bound policy
      kernels of one half beside the issue-bound physics of the other),
  (c) the two half handles one after the other on ONE stream (what the split alone costs).
Usage: python tools/dev/two_stream_probe.py [task] [num_envs] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-quadruped-environment_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import make_desc, hip_engine  # noqa: E402
from mqe.engine import abi  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "go1gate"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300


def make(n):
    d, keep, _ = make_desc(task, n)
    e = hip_engine(d, keep)
    e.reset_all()
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = [torch.rand(n, Aw, 3, device="cuda", generator=g) * 2 - 1 for _ in range(16)]
    for t in range(40):
        e.step(acts[t % 16])
    return e, acts


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


whole, aw = make(N)
h1, a1 = make(N // 2)
h2, a2 = make(N // 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def both_streams(i):
    with torch.cuda.stream(s1):
        h1.step(a1[i % 16])
    with torch.cuda.stream(s2):
        h2.step(a2[i % 16])


def one_stream(i):
    h1.step(a1[i % 16])
    h2.step(a2[i % 16])


t_whole = timed(lambda i: whole.step(aw[i % 16]), steps)
t_two = timed(both_streams, steps)
t_seq = timed(one_stream, steps)
t_whole2 = timed(lambda i: whole.step(aw[i % 16]), steps)
t_half = timed(lambda i: h1.step(a1[i % 16]), steps)
print(f"{task}: one handle of {N} envs {t_whole:.4f} / {t_whole2:.4f} ms per step; two handles of {N // 2} on two streams {t_two:.4f}; the same two on one stream {t_seq:.4f}; "
      f"one handle of {N // 2} alone {t_half:.4f}")
