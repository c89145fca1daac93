import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import torch
import bench
for envs in (64, 256):
    t0 = time.time()
    v, secs, nthr = bench.cpu_baseline("go1gate", envs, 6)
    print(os.environ.get("OMP_NUM_THREADS"), envs, round(v, 1), round(secs, 2), nthr, flush=True)
